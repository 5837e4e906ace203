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

// grid (ceil(w / kSppPix), h, N), block = kSppPix pixels x ctot4 16-byte words (320 threads at the path's 64 + 128 + 4 x 32 channels),
// ordered BY SOURCE: first the strip's quarter words, then its deep words, then its branch words — at the path's widths every wave
// takes ONE of the three code paths (64 = 4 pixels x 16 quarter words, 128 deep, 128 branch).  Round 6: the first version walked a flat
// index (pixel, word), so most waves ran all three paths one after the other, each with its own memory latency in front of the one
// store: 2.4 TB/s; and it divided a 64-bit index three times per thread.
constexpr int kSppPix = 4;

constexpr int kSppRows = 2;      // image rows per workgroup: a thread keeps the loads of both rows in flight before it stores

__global__ __launch_bounds__(512) void spp_concat_kernel(const SppArgs a) {
    const int cq4 = a.Cq >> 2, cd4 = a.Cd >> 2, cb4 = a.Cb >> 2, ctot4 = cq4 + cd4 + 4 * cb4;
    const int tid = threadIdx.x;
    const int nq = kSppPix * cq4, nd = kSppPix * cd4, nb = kSppPix * 4 * cb4;
    if (tid >= nq + nd + nb) return;
    // (pixel of the strip, word inside the source): compare instead of dividing by a run-time width
    auto split = [&](int t, int per, int& pl, int& g) {
        pl = 0; g = t;
#pragma unroll
        for (int i = 1; i < kSppPix; ++i)
            if (t >= i * per) { pl = i; g = t - i * per; }
    };
    const int y0 = blockIdx.y * kSppRows, n = blockIdx.z;
    const int nrow = min(kSppRows, a.h - y0);
    int pl, g;
    sf32x4 v[kSppRows];
    if (tid < nq + nd) {
        const bool q = tid < nq;
        split(q ? tid : tid - nq, q ? cq4 : cd4, pl, g);
        const int x = blockIdx.x * kSppPix + pl;
        if (x >= a.w) return;
        const sf32x4* src = reinterpret_cast<const sf32x4*>(q ? a.quarter : a.deep);
        const int c4n = q ? cq4 : cd4, goff = q ? g : cq4 + g;
        const size_t pix = ((size_t)n * a.h + y0) * a.w + x;
#pragma unroll
        for (int r = 0; r < kSppRows; ++r)
            if (r < nrow) v[r] = src[(pix + (size_t)r * a.w) * c4n + g];
#pragma unroll
        for (int r = 0; r < kSppRows; ++r)
            if (r < nrow) reinterpret_cast<sf32x4*>(a.out)[(pix + (size_t)r * a.w) * ctot4 + goff] = v[r];
        return;
    }
    split(tid - nq - nd, 4 * cb4, pl, g);
    const int x = blockIdx.x * kSppPix + pl;
    if (x >= a.w) return;
    const size_t pix = ((size_t)n * a.h + y0) * a.w + x;
    int gb = g, b = 0;
#pragma unroll
    for (int i = 1; i < 4; ++i)
        if (gb >= cb4) { gb -= cb4; b = i; }
    const int c4 = gb;
    const int bh = a.bh[b], bw = a.bw[b];
    const float sch = a.h > 1 ? (float)(bh - 1) / (float)(a.h - 1) : 0.f;
    const float scw = a.w > 1 ? (float)(bw - 1) / (float)(a.w - 1) : 0.f;
    const float w1r = scw * (float)x;
    const int w1 = min((int)w1r, bw - 1), w1p = w1 < bw - 1 ? 1 : 0;
    const float w1l = fminf(fmaxf(w1r - (float)w1, 0.f), 1.f), w0l = 1.f - w1l;
    const sf32x4* z = reinterpret_cast<const sf32x4*>(a.bz[b]) + ((size_t)n * bh * bw) * cb4 + c4;
    const float* ss = a.bss[b] + 8 * c4;                   // (scale, shift) of channels 4 c4 .. 4 c4 + 3
    const sf32x4 sc = {ss[0], ss[2], ss[4], ss[6]}, sh = {ss[1], ss[3], ss[5], ss[7]};
    auto act = [&](sf32x4 r) -> sf32x4 {
        r = r * sc + sh;                                    // BatchNorm (batch statistics) of the tiny map ...
        r.x = fmaxf(r.x, 0.f); r.y = fmaxf(r.y, 0.f); r.z = fmaxf(r.z, 0.f); r.w = fmaxf(r.w, 0.f);   // ... and its ReLU
        return r;
    };
    // the taps of both rows are requested together, then normalised and blended: one memory latency per thread
    sf32x4 r00[kSppRows], r01[kSppRows], r10[kSppRows], r11[kSppRows];
    float h0l[kSppRows], h1l[kSppRows];
#pragma unroll
    for (int r = 0; r < kSppRows; ++r) {
        const int y = min(y0 + r, a.h - 1);
        const float h1r = sch * (float)y;
        const int h1 = min((int)h1r, bh - 1), h1p = h1 < bh - 1 ? 1 : 0;
        h1l[r] = fminf(fmaxf(h1r - (float)h1, 0.f), 1.f); h0l[r] = 1.f - h1l[r];
        r00[r] = z[(h1 * bw + w1) * cb4]; r01[r] = z[(h1 * bw + w1 + w1p) * cb4];
        r10[r] = z[((h1 + h1p) * bw + w1) * cb4]; r11[r] = z[((h1 + h1p) * bw + w1 + w1p) * cb4];
    }
#pragma unroll
    for (int r = 0; r < kSppRows; ++r) {
        const sf32x4 v00 = act(r00[r]), v01 = act(r01[r]), v10 = act(r10[r]), v11 = act(r11[r]);
        v[r] = h0l[r] * (w0l * v00 + w1l * v01) + h1l[r] * (w0l * v10 + w1l * v11);
        if (r < nrow) reinterpret_cast<sf32x4*>(a.out)[(pix + (size_t)r * a.w) * ctot4 + cq4 + cd4 + g] = v[r];
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
    const int ctot4 = (Cq + Cd + 4 * Cb) >> 2;
    if (kSppPix * ctot4 > 512 || h > 65535 || N > 65535) return NRGBD_E_SHAPE;        // one workgroup = kSppPix pixels of a row
    const int threads = (kSppPix * ctot4 + 63) / 64 * 64;
    hipLaunchKernelGGL(spp_concat_kernel, dim3((unsigned)((w + kSppPix - 1) / kSppPix), (unsigned)((h + kSppRows - 1) / kSppRows), (unsigned)N), dim3(threads), 0,
                       (hipStream_t)stream, a);
    NRGBD_CHECK_LAUNCH();
    return NRGBD_OK;
}
