// warpvol.hip — plane-sweep warp of low-channel maps with the samples kept + K-Net input volume.
// Replaces warping/homography.py:234-280 (warp_img_feats_v3), :183-232 (warp_img_feats_mgpu) and
// the torch.cat of models/KVNET.py:163-166.  HBM-write-bound: (V*Cs + Cs + 1) * D * hw floats are
// written once, coalesced along x; the small source maps stay in L2.
#include "common.hpp"

namespace nrgbd {

struct WarpVolArgs {
    const float* src; long sv, sc, sy, sx;
    const float* ref; long rc, ry, rx;
    const float* KR; const float* Kt; const float* rays; const float* d_candi;
    const float* bv_cur; const float* bv_pred;
    float* out;
    float cx, cy;
    int align, V, Cs, D, h, w;
    int channels_last;  // 0: out [Ch][D][h][w] (torch NCDHW), 1: out [D][h][w][Ch] (conv3d.hip input layout)
};

// grid: (ceil(hw/256), D); one lane = one (pixel, depth) pair, loops views and channels.
__global__ __launch_bounds__(256) void warp_volume_kernel(const WarpVolArgs a) {
    const size_t hw = (size_t)a.h * a.w;
    const size_t p = (size_t)blockIdx.x * 256 + threadIdx.x;
    const int k = blockIdx.y;
    if (p >= hw) return;
    const int y = (int)(p / a.w), x = (int)(p - (size_t)y * a.w);
    const float rx = a.rays[p], ry = a.rays[hw + p], rz = a.rays[2 * hw + p];
    const float d = a.d_candi[k];
    const float wf = (float)a.w, hf = (float)a.h;
    const int n_ch = a.V * a.Cs + (a.ref ? a.Cs : 0) + (a.bv_cur ? 1 : 0);
    // element (channel ch, depth k, pixel p) lives at o[ch * plane]
    const size_t plane = a.channels_last ? 1 : (size_t)a.D * hw;
    float* o = a.channels_last ? a.out + ((size_t)k * hw + p) * n_ch : a.out + (size_t)k * hw + p;
    for (int v = 0; v < a.V; ++v) {
        const SweepTerm st = make_sweep_term(a.KR + 9 * v, a.Kt + 3 * v, rx, ry, rz);
        float ix, iy;
        sweep_sample_pos(st, d, a.cx, a.cy, wf, hf, a.align != 0, ix, iy);
        const Bilinear b = bilinear_zeros(ix, iy, a.w, a.h);
        const float* s = a.src + v * a.sv;
        const long onw = b.y0 * a.sy + b.x0 * a.sx, one = b.y0 * a.sy + b.x1 * a.sx;
        const long osw = b.y1 * a.sy + b.x0 * a.sx, ose = b.y1 * a.sy + b.x1 * a.sx;
        for (int c = 0; c < a.Cs; ++c) {
            const float* pl = s + c * a.sc;
            o[(size_t)(v * a.Cs + c) * plane] = lerp4(pl[onw], pl[one], pl[osw], pl[ose], b);
        }
    }
    int ch = a.V * a.Cs;
    if (a.ref) {  // reference map repeated over D (KVNET.py:163)
        for (int c = 0; c < a.Cs; ++c) o[(size_t)(ch + c) * plane] = a.ref[c * a.rc + y * a.ry + x * a.rx];
        ch += a.Cs;
    }
    if (a.bv_cur) o[(size_t)ch * plane] = a.bv_cur[(size_t)k * hw + p] - a.bv_pred[(size_t)k * hw + p];
}

// K-Net input assembly specialised for the layout the model uses: sources and reference are the RGB word
// (channels 64..67 = R,G,B,0: one aligned 16-B word) of the NHWC texel tensor, output channels-last
// [D][h][w][16] for conv3d.hip.  One lane = one (pixel, depth): 4 views x 4 taps x one 16-B load, the
// 16 output channels leave as four 16-B stores (64 contiguous bytes per lane).
template <bool ALIGN>
__global__ __launch_bounds__(256) void warp_volume_cl16_kernel(const WarpVolArgs a, float rcx, float rcy) {
    const size_t hw = (size_t)a.h * a.w;
    const size_t p = (size_t)blockIdx.x * 256 + threadIdx.x;
    const int k = blockIdx.y;
    if (p >= hw) return;
    const int y = (int)(p / a.w), x = (int)(p - (size_t)y * a.w);
    const float rx = a.rays[p], ry = a.rays[hw + p], rz = a.rays[2 * hw + p];
    const float d = a.d_candi[k];
    const float wf = (float)a.w, hf = (float)a.h;
    float o[16];
#pragma unroll
    for (int v = 0; v < 4; ++v) {
        const SweepTerm st = make_sweep_term(a.KR + 9 * v, a.Kt + 3 * v, rx, ry, rz);
        float ix, iy;
        // same bits as sweep_sample_pos (common.hpp: exact constant-divisor and shared-reciprocal divisions)
        sweep_sample_pos_fast<ALIGN>(st, d, a.cx, a.cy, rcx, rcy, wf, hf, ix, iy);
        const Bilinear b = bilinear_zeros(ix, iy, a.w, a.h);
        const float* s = a.src + v * a.sv;
        const float4 A = *reinterpret_cast<const float4*>(s + b.y0 * a.sy + b.x0 * a.sx);
        const float4 B = *reinterpret_cast<const float4*>(s + b.y0 * a.sy + b.x1 * a.sx);
        const float4 C = *reinterpret_cast<const float4*>(s + b.y1 * a.sy + b.x0 * a.sx);
        const float4 Dd = *reinterpret_cast<const float4*>(s + b.y1 * a.sy + b.x1 * a.sx);
        o[3 * v + 0] = lerp4(A.x, B.x, C.x, Dd.x, b);
        o[3 * v + 1] = lerp4(A.y, B.y, C.y, Dd.y, b);
        o[3 * v + 2] = lerp4(A.z, B.z, C.z, Dd.z, b);
    }
    const float4 r = *reinterpret_cast<const float4*>(a.ref + y * a.ry + x * a.rx);
    o[12] = r.x; o[13] = r.y; o[14] = r.z;
    o[15] = a.bv_cur[(size_t)k * hw + p] - a.bv_pred[(size_t)k * hw + p];
    float4* dst = reinterpret_cast<float4*>(a.out + ((size_t)k * hw + p) * 16);
#pragma unroll
    for (int q = 0; q < 4; ++q) dst[q] = make_float4(o[4 * q], o[4 * q + 1], o[4 * q + 2], o[4 * q + 3]);
}

}  // namespace nrgbd

extern "C" int nrgbd_warp_volume(const float* src, long sv, long sc, long sy, long sx,
                                 const float* ref, long rc, long ry, long rx, const float* KR,
                                 const float* Kt, const float* rays, const float* d_candi,
                                 float cx, float cy, int align_corners, const float* bv_cur,
                                 const float* bv_pred, float* out, int V, int Cs, int D, int h,
                                 int w, int channels_last, void* stream) {
    using namespace nrgbd;
    if (!src || !KR || !Kt || !rays || !d_candi || !out) return NRGBD_E_NULL;
    if ((bv_cur == nullptr) != (bv_pred == nullptr)) return NRGBD_E_NULL;
    if (V <= 0 || V > NRGBD_MAX_V || Cs <= 0 || D <= 0 || D > 65535 || h <= 0 || w <= 0) return NRGBD_E_SHAPE;
    WarpVolArgs a{src, sv, sc, sy, sx, ref, rc, ry, rx, KR, Kt, rays, d_candi, bv_cur, bv_pred,
                  out, cx, cy, align_corners, V, Cs, D, h, w, channels_last};
    dim3 grid(ceil_div((long)h * w, 256), D);
    const bool word_src = Cs == 3 && sc == 1 && rc == 1 && !((sv | sy | sx | ry | rx) & 3) &&
                          !((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(ref) |
                             reinterpret_cast<uintptr_t>(out)) & 15);
    const float rcx = (float)(1.0 / (double)cx), rcy = (float)(1.0 / (double)cy);
    if (channels_last && V == 4 && ref && bv_cur && word_src) {
        if (align_corners) hipLaunchKernelGGL(warp_volume_cl16_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, a, rcx, rcy);
        else hipLaunchKernelGGL(warp_volume_cl16_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, a, rcx, rcy);
    } else
        hipLaunchKernelGGL(warp_volume_kernel, grid, dim3(256), 0, (hipStream_t)stream, a);
    NRGBD_CHECK_LAUNCH();
    return NRGBD_OK;
}
