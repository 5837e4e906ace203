// costvol_bwd.hip — backward of the fused plane-sweep cost volume (training, BASELINE config 4).
// Replaces what autograd does for warping/homography.py:293-331 in the reference: grid_sample backward
// (a scatter-add of the bilinear weights into the source features), the broadcast subtraction and the
// channel sum.  Nothing of the [D,C,h,w] warped tensors is materialised in either direction.
//
//   cost[k,p] = sum_v (1/sigma) sum_c dist( s_c(v,k,p) - ref_c(p) ),   s_c = sum_tap w_tap * src_v[c, tap]
//   L2:  dcost/ds_c = 2 (s_c - ref_c) / sigma      L1:  sign(s_c - ref_c) / sigma
//   g_src[v, tap, c] += w_tap * dcost/ds_c * g[k,p]        (atomic: many pixels/candidates hit one texel)
//   g_ref[p, c]      -= sum_{k,v} dcost/ds_c * g[k,p]      (register accumulation, one atomic per word)
// Sampling positions are recomputed exactly as in the forward kernels (no gradient flows to the poses:
// they are inputs, as in the reference where they come from the dataset).
#include "costvol.hpp"

namespace nrgbd {

struct CostvolBwdArgs {
    const float* ref; const float* src; const float* KR; const float* Kt; const float* rays;
    const float* d_candi; const float* g_cost;
    float* g_ref; float* g_src;
    float cx, cy, sigma;
    int dist, align, V, C, Cp, D, h, w, kchunks;
};

// grid (ceil(hw/64), kchunks, Cp/4), block 64: one lane = one pixel x one slice of consecutive depth candidates x
// one 16-byte channel word.  Consecutive candidates of a pixel sample neighbouring positions along its epipolar
// line — for the far planes the SAME 2x2 source cell for several candidates in a row — so the four tap gradients
// are accumulated in registers while the cell stays the same and flushed with atomics only when it changes
// (4-8x fewer atomics than one per (pixel, candidate, view, channel); the kernel is atomic-bound).
__global__ __launch_bounds__(64) void costvol_bwd_kernel(const CostvolBwdArgs a) {
    const size_t hw = (size_t)a.h * a.w;
    const size_t p = (size_t)blockIdx.x * 64 + threadIdx.x;
    if (p >= hw) return;
    const int per = (a.D + a.kchunks - 1) / a.kchunks;
    const int k_begin = blockIdx.y * per, k_end = min(a.D, k_begin + per);
    const int i = blockIdx.z;                      // channel word
    const float rx = a.rays[p], ry = a.rays[hw + p], rz = a.rays[2 * hw + p];
    const float wf = (float)a.w, hf = (float)a.h;
    const float4 r = *reinterpret_cast<const float4*>(a.ref + p * a.Cp + 4 * i);
    const int ncomp = min(4, a.C - 4 * i);         // valid components of this word
    float gr[4] = {0.f, 0.f, 0.f, 0.f};

    for (int v = 0; v < a.V; ++v) {
        const SweepTerm st = make_sweep_term(a.KR + 9 * v, a.Kt + 3 * v, rx, ry, rz);
        const float* sv = a.src + (size_t)v * hw * a.Cp + 4 * i;
        float* gs = a.g_src + (size_t)v * hw * a.Cp + 4 * i;
        // register accumulator of the current cell: 4 taps x 4 components
        float acc[4][4];
        float cx0 = -1e30f, cy0 = -1e30f;          // floor of the current cell (never matches initially)
        size_t o[4] = {0, 0, 0, 0};
        bool have = false;
        auto flush = [&]() {
            if (!have) return;
#pragma unroll
            for (int tpi = 0; tpi < 4; ++tpi)
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (acc[tpi][e] != 0.f) atomicAdd(gs + o[tpi] + e, acc[tpi][e]);
        };
        for (int k = k_begin; k < k_end; ++k) {
            const float gk = a.g_cost[(size_t)k * hw + p] / a.sigma;
            float ix, iy;
            sweep_sample_pos(st, a.d_candi[k], a.cx, a.cy, wf, hf, a.align != 0, ix, iy);
            const Bilinear b = bilinear_zeros(ix, iy, a.w, a.h);
            const float x0f = floorf(ix), y0f = floorf(iy);
            if (!(x0f == cx0 && y0f == cy0)) {      // new cell (also taken for NaN positions)
                flush();
                cx0 = x0f; cy0 = y0f; have = true;
                o[0] = ((size_t)b.y0 * a.w + b.x0) * a.Cp; o[1] = ((size_t)b.y0 * a.w + b.x1) * a.Cp;
                o[2] = ((size_t)b.y1 * a.w + b.x0) * a.Cp; o[3] = ((size_t)b.y1 * a.w + b.x1) * a.Cp;
#pragma unroll
                for (int tpi = 0; tpi < 4; ++tpi)
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[tpi][e] = 0.f;
            }
            if (gk == 0.f) continue;
            const float4 A = *reinterpret_cast<const float4*>(sv + o[0]), B = *reinterpret_cast<const float4*>(sv + o[1]);
            const float4 Cc = *reinterpret_cast<const float4*>(sv + o[2]), Dd = *reinterpret_cast<const float4*>(sv + o[3]);
            const float df[4] = {lerp4(A.x, B.x, Cc.x, Dd.x, b) - r.x, lerp4(A.y, B.y, Cc.y, Dd.y, b) - r.y,
                                 lerp4(A.z, B.z, Cc.z, Dd.z, b) - r.z, lerp4(A.w, B.w, Cc.w, Dd.w, b) - r.w};
            const float wt[4] = {b.nw, b.ne, b.sw, b.se};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if (e >= ncomp) continue;
                const float ds = (a.dist == NRGBD_DIST_L2) ? 2.f * df[e] : (df[e] > 0.f ? 1.f : (df[e] < 0.f ? -1.f : 0.f));
                const float c = ds * gk;
                gr[e] -= c;
#pragma unroll
                for (int tpi = 0; tpi < 4; ++tpi) acc[tpi][e] = __builtin_fmaf(wt[tpi], c, acc[tpi][e]);
            }
        }
        flush();
    }
    float* go = a.g_ref + p * a.Cp + 4 * i;
#pragma unroll
    for (int e = 0; e < 4; ++e)
        if (gr[e] != 0.f) atomicAdd(go + e, gr[e]);
}

}  // namespace nrgbd

extern "C" int nrgbd_costvol_bwd(const float* ref_nhwc, const float* src_nhwc, const float* KR, const float* Kt,
                                 const float* rays, const float* d_candi, float cx, float cy, float sigma,
                                 int dist, int align_corners, const float* g_cost, float* g_ref, float* g_src,
                                 int V, int C, int Cp, int D, int h, int w, void* stream) {
    using namespace nrgbd;
    if (!ref_nhwc || !src_nhwc || !KR || !Kt || !rays || !d_candi || !g_cost || !g_ref || !g_src) return NRGBD_E_NULL;
    if (V <= 0 || V > NRGBD_MAX_V || C <= 0 || D <= 0 || D > NRGBD_MAX_D || h <= 0 || w <= 0) return NRGBD_E_SHAPE;
    if ((Cp & 3) || Cp < C || Cp - C > 3) return NRGBD_E_ALIGN;
    if (dist != NRGBD_DIST_L2 && dist != NRGBD_DIST_L1) return NRGBD_E_ARG;
    hipStream_t s = (hipStream_t)stream;
    const size_t hw = (size_t)h * w;
    hipError_t e = hipMemsetAsync(g_ref, 0, hw * Cp * sizeof(float), s);
    if (e != hipSuccess) return (int)e;
    e = hipMemsetAsync(g_src, 0, (size_t)V * hw * Cp * sizeof(float), s);
    if (e != hipSuccess) return (int)e;
    // slices of consecutive candidates: long enough for the register run-accumulation, parallelism comes from
    // the channel-word axis of the grid
    const int kchunks = D >= 32 ? 4 : (D >= 8 ? 2 : 1);
    CostvolBwdArgs a{ref_nhwc, src_nhwc, KR, Kt, rays, d_candi, g_cost, g_ref, g_src, cx, cy, sigma,
                     dist, align_corners, V, C, Cp, D, h, w, kchunks};
    hipLaunchKernelGGL(costvol_bwd_kernel, dim3(ceil_div((long)hw, 64), kchunks, Cp >> 2), dim3(64), 0, s, a);
    NRGBD_CHECK_LAUNCH();
    return NRGBD_OK;
}
