// spp.hip — the tail of the feature CNN's spatial-pyramid pooling (models/psm_submodule.py:149-161) as ONE channels-last pass:
// for every pixel of the quarter-resolution grid the 320-channel input of `lastconv` is assembled in place —
//     [ quarter (layer2, 64) | deep (layer4, 128) | branch4 | branch3 | branch2 | branch1 (4 x 32) ]
// where branch_i = bilinear up-sampling (align_corners=True, :153-158 F.upsample) of relu(bn(conv1x1(avg_pool_i(deep)))) from its
// tiny map (24x32 ... 3x4 at config B).  The BatchNorm + ReLU of the four tiny maps is applied at the taps (scale, shift per
// channel), so this replaces four nhwc_act passes, four upsample_bilinear2d launches and the CatArrayBatchedCopy (three launches)
// of the torch glue: 503 MB of traffic in one HBM-bound launch instead of 11.
// Interpolation arithmetic = ATen's upsample_bilinear2d: scale = (in - 1) / (out - 1) in fp32, source = scale * dst, lower index =
// min((int)source, in - 1), upper = lower + (lower < in - 1), lambda1 = source - lower,
// value = h0 (w0 v00 + w1 v01) + h1 (w0 v10 + w1 v11), every operation rounded once (-ffp-contract=off).
#include "common.hpp"

namespace nrgbd {

struct SppArgs {
    const float* quarter; const float* deep;
    const float* bz[4]; const float* bss[4];    // branch i: raw 1x1-conv output [N][bh][bw][Cb] and its (scale, shift) [Cb][2]
    int bh[4], bw[4];
    float* out;
    int N, h, w, Cq, Cd, Cb;
};

typedef float sf32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void spp_concat_kernel(const SppArgs a) {
    const int cq4 = a.Cq >> 2, cd4 = a.Cd >> 2, cb4 = a.Cb >> 2, ctot4 = cq4 + cd4 + 4 * cb4;
    const long total = (long)a.N * a.h * a.w * ctot4;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const long pix = idx / ctot4;
        const int g = (int)(idx - pix * ctot4);
        sf32x4 v;
        if (g < cq4) {
            v = reinterpret_cast<const sf32x4*>(a.quarter)[pix * cq4 + g];
        } else if (g < cq4 + cd4) {
            v = reinterpret_cast<const sf32x4*>(a.deep)[pix * cd4 + (g - cq4)];
        } else {
            const int gb = g - cq4 - cd4, b = gb / cb4, c4 = gb - b * cb4;
            const int x = (int)(pix % a.w);
            const long t = pix / a.w;
            const int y = (int)(t % a.h), n = (int)(t / a.h);
            const int bh = a.bh[b], bw = a.bw[b];
            const float sch = a.h > 1 ? (float)(bh - 1) / (float)(a.h - 1) : 0.f;
            const float scw = a.w > 1 ? (float)(bw - 1) / (float)(a.w - 1) : 0.f;
            const float h1r = sch * (float)y, w1r = scw * (float)x;
            const int h1 = min((int)h1r, bh - 1), w1 = min((int)w1r, bw - 1);
            const int h1p = h1 < bh - 1 ? 1 : 0, w1p = w1 < bw - 1 ? 1 : 0;
            const float h1l = fminf(fmaxf(h1r - (float)h1, 0.f), 1.f), h0l = 1.f - h1l;
            const float w1l = fminf(fmaxf(w1r - (float)w1, 0.f), 1.f), w0l = 1.f - w1l;
            const sf32x4* z = reinterpret_cast<const sf32x4*>(a.bz[b]) + ((long)n * bh * bw) * cb4 + c4;
            const float* ss = a.bss[b] + 8 * c4;               // (scale, shift) of channels 4 c4 .. 4 c4 + 3
            const sf32x4 sc = {ss[0], ss[2], ss[4], ss[6]}, sh = {ss[1], ss[3], ss[5], ss[7]};
            auto tap = [&](int yy, int xx) -> sf32x4 {
                sf32x4 r = z[((long)yy * bw + xx) * cb4] * sc + sh;   // BatchNorm (batch statistics) of the tiny map ...
                r.x = fmaxf(r.x, 0.f); r.y = fmaxf(r.y, 0.f); r.z = fmaxf(r.z, 0.f); r.w = fmaxf(r.w, 0.f);   // ... and its ReLU
                return r;
            };
            const sf32x4 v00 = tap(h1, w1), v01 = tap(h1, w1 + w1p), v10 = tap(h1 + h1p, w1), v11 = tap(h1 + h1p, w1 + w1p);
            v = h0l * (w0l * v00 + w1l * v01) + h1l * (w0l * v10 + w1l * v11);
        }
        reinterpret_cast<sf32x4*>(a.out)[idx] = v;
    }
}

// ---- training: the bilinear up-sampling (align_corners = True) of one tiny SPP map and its exact adjoint, channels-last.
// The weights of an output row / column are ATen's (see the header comment); the backward thread of an INPUT element walks the
// output rows / columns that can reach it and recomputes the same fp32 weights, so it is the adjoint of the forward to the bit
// and sums in a fixed order (F.interpolate's backward scatters with atomics: 0.82 ms per branch at the ScanNet grid and not
// reproducible; the two-matmul form of rounds 2-3 ran on rocBLAS).
struct UpW { int i0, i1; float l0, l1; };
__device__ __forceinline__ UpW up_weights(int dst, int n_in, float scale) {
    const float r = scale * (float)dst;
    UpW u;
    u.i0 = min((int)r, n_in - 1);
    u.i1 = u.i0 + (u.i0 < n_in - 1 ? 1 : 0);
    u.l1 = fminf(fmaxf(r - (float)u.i0, 0.f), 1.f);
    u.l0 = 1.f - u.l1;
    return u;
}

__global__ __launch_bounds__(256) void upsample_ac_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int N, int bh, int bw,
                                                              int H, int W, int C) {
    const int c4n = C >> 2;
    const long total = (long)N * H * W * c4n;
    const float sch = H > 1 ? (float)(bh - 1) / (float)(H - 1) : 0.f, scw = W > 1 ? (float)(bw - 1) / (float)(W - 1) : 0.f;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const int c4 = (int)(idx % c4n);
        long t = idx / c4n;
        const int xo = (int)(t % W); t /= W;
        const int yo = (int)(t % H);
        const int n = (int)(t / H);
        const UpW a = up_weights(yo, bh, sch), b = up_weights(xo, bw, scw);
        const sf32x4* z = reinterpret_cast<const sf32x4*>(x) + (long)n * bh * bw * c4n + c4;
        const sf32x4 v00 = z[((long)a.i0 * bw + b.i0) * c4n], v01 = z[((long)a.i0 * bw + b.i1) * c4n];
        const sf32x4 v10 = z[((long)a.i1 * bw + b.i0) * c4n], v11 = z[((long)a.i1 * bw + b.i1) * c4n];
        reinterpret_cast<sf32x4*>(y)[idx] = a.l0 * (b.l0 * v00 + b.l1 * v01) + a.l1 * (b.l0 * v10 + b.l1 * v11);
    }
}

// one workgroup per INPUT pixel (n, by, bx): thread = (16-byte channel word c4, slice pg of the support box); the slices walk the
// box with stride 256 / (C / 4) and are combined by a fixed-order tree in LDS — the coarsest SPP map (1 x 1 for a 64 x 96 grid)
// has the whole output as its support, which one thread per element would sum serially (measured: 2.3 ms)
__global__ __launch_bounds__(256) void upsample_ac_bwd_kernel(const float* __restrict__ gy, float* __restrict__ gx, int N, int bh, int bw,
                                                              int H, int W, int C) {
    __shared__ sf32x4 part[256];
    const int c4n = C >> 2, tid = threadIdx.x;
    const int npg = 256 / c4n;                              // c4n <= 64 (C <= 256): >= 4 slices
    const int c4 = tid % c4n, pg = tid / c4n;
    const float sch = H > 1 ? (float)(bh - 1) / (float)(H - 1) : 0.f, scw = W > 1 ? (float)(bw - 1) / (float)(W - 1) : 0.f;
    int t = blockIdx.x;
    const int bx = t % bw; t /= bw;
    const int by = t % bh;
    const int n = t / bh;
    // output rows / columns whose source position lies in (b - 1, b + 1), with a margin of one for the fp32 rounding of
    // scale * dst; membership is then decided by the forward's own weights
    int y0 = 0, y1 = H - 1, x0 = 0, x1 = W - 1;
    if (sch > 0.f) { y0 = max(0, (int)floorf((float)(by - 1) / sch) - 1); y1 = min(H - 1, (int)ceilf((float)(by + 1) / sch) + 1); }
    if (scw > 0.f) { x0 = max(0, (int)floorf((float)(bx - 1) / scw) - 1); x1 = min(W - 1, (int)ceilf((float)(bx + 1) / scw) + 1); }
    const int bwid = x1 - x0 + 1, box = bwid * (y1 - y0 + 1);
    const sf32x4* g = reinterpret_cast<const sf32x4*>(gy) + (long)n * H * W * c4n + c4;
    sf32x4 acc = {0.f, 0.f, 0.f, 0.f};
    if (pg < npg) {
        for (int q = pg; q < box; q += npg) {
            const int yo = y0 + q / bwid, xo = x0 + q % bwid;
            const UpW a = up_weights(yo, bh, sch), b = up_weights(xo, bw, scw);
            const float wy = (a.i0 == by ? a.l0 : 0.f) + (a.i1 == by ? a.l1 : 0.f);
            const float wx = (b.i0 == bx ? b.l0 : 0.f) + (b.i1 == bx ? b.l1 : 0.f);
            const float wgt = wy * wx;
            if (wgt != 0.f) acc = acc + wgt * g[((long)yo * W + xo) * c4n];
        }
    }
    part[tid] = acc;
    __syncthreads();
    for (int s = 1; s < npg; s <<= 1) {                     // slice pg accumulates slice pg + s: the same tree for every launch
        if (pg < npg && (pg & (2 * s - 1)) == 0 && pg + s < npg) part[tid] = part[tid] + part[tid + s * c4n];
        __syncthreads();
    }
    if (pg == 0) reinterpret_cast<sf32x4*>(gx)[((long)blockIdx.x) * c4n + c4] = part[tid];
}

}  // namespace nrgbd

extern "C" int nrgbd_upsample_bilinear_ac(const float* x, float* y, int N, int bh, int bw, int H, int W, int C, int backward,
                                          void* stream) {
    using namespace nrgbd;
    if (!x || !y) return NRGBD_E_NULL;
    if (N <= 0 || bh <= 0 || bw <= 0 || H <= 0 || W <= 0 || C <= 0) return NRGBD_E_SHAPE;
    if (C & 3) return NRGBD_E_ALIGN;
    if (backward) {
        if (C > 256 || (long)N * bh * bw >= (1L << 31)) return NRGBD_E_SHAPE;       // one workgroup per input pixel, <= 64 channel words
        hipLaunchKernelGGL(upsample_ac_bwd_kernel, dim3((unsigned)((long)N * bh * bw)), dim3(256), 0, (hipStream_t)stream, x, y, N, bh, bw, H, W, C);
    } else {
        const long blocks = ((long)N * H * W * (C >> 2) + 255) / 256;
        hipLaunchKernelGGL(upsample_ac_fwd_kernel, dim3((unsigned)(blocks < 8192 ? blocks : 8192)), dim3(256), 0, (hipStream_t)stream, x, y, N, bh, bw, H, W, C);
    }
    NRGBD_CHECK_LAUNCH();
    return NRGBD_OK;
}

extern "C" int nrgbd_spp_concat(const float* quarter, int Cq, const float* deep, int Cd,
                                const float* bz0, const float* bss0, int bh0, int bw0,
                                const float* bz1, const float* bss1, int bh1, int bw1,
                                const float* bz2, const float* bss2, int bh2, int bw2,
                                const float* bz3, const float* bss3, int bh3, int bw3,
                                int Cb, float* out, int N, int h, int w, void* stream) {
    using namespace nrgbd;
    if (!quarter || !deep || !bz0 || !bz1 || !bz2 || !bz3 || !bss0 || !bss1 || !bss2 || !bss3 || !out) return NRGBD_E_NULL;
    if (N <= 0 || h <= 0 || w <= 0 || Cq <= 0 || Cd <= 0 || Cb <= 0 || bh0 <= 0 || bh1 <= 0 || bh2 <= 0 || bh3 <= 0 ||
        bw0 <= 0 || bw1 <= 0 || bw2 <= 0 || bw3 <= 0) return NRGBD_E_SHAPE;
    if ((Cq | Cd | Cb) & 3) return NRGBD_E_ALIGN;
    SppArgs a{quarter, deep, {bz0, bz1, bz2, bz3}, {bss0, bss1, bss2, bss3}, {bh0, bh1, bh2, bh3}, {bw0, bw1, bw2, bw3}, out,
              N, h, w, Cq, Cd, Cb};
    const long total = (long)N * h * w * ((Cq + Cd + 4 * Cb) >> 2);
    const long blocks = (total + 255) / 256;
    hipLaunchKernelGGL(spp_concat_kernel, dim3((unsigned)(blocks < 8192 ? blocks : 8192)), dim3(256), 0, (hipStream_t)stream, a);
    NRGBD_CHECK_LAUNCH();
    return NRGBD_OK;
}
