// costvol_bwd.hip — backward of the fused plane-sweep cost volume (training, BASELINE config 4).
// Replaces what autograd does for warping/homography.py:293-331 in the reference: grid_sample backward
// (a scatter-add of the bilinear weights into the source features), the broadcast subtraction and the
// channel sum.  Nothing of the [D,C,h,w] warped tensors is materialised in either direction.
//
//   cost[k,p] = sum_v (1/sigma) sum_c dist( s_c(v,k,p) - ref_c(p) ),   s_c = sum_tap w_tap * src_v[c, tap]
//   L2:  dcost/ds_c = 2 (s_c - ref_c) / sigma      L1:  sign(s_c - ref_c) / sigma
//   g_src[v, tap, c] += w_tap * dcost/ds_c * g[k,p]        (scatter: many pixels/candidates hit one texel)
//   g_ref[p, c]      -= sum_{k,v} dcost/ds_c * g[k,p]      (register accumulation)
// Two kernels.  costvol_bwd_lds_kernel (grids whose 16-byte channel word of one source view fits in LDS, i.e. the
// training grids): the scatter goes into LDS with ds_add_f32 and leaves the workgroup once, as plain coalesced
// stores of per-slice partial sums that costvol_bwd_reduce_kernel adds in a fixed order — no global atomics at
// all.  costvol_bwd_kernel (larger grids): the same sample loop with global atomics; measured on MI355X at
// 64x96x64, V = 4, C = 67 it spends 4.9 of its 5.2 ms waiting on those atomics, the LDS kernel takes the rest.
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
    int abl;   // developer builds only (NRGBD_BWD_ABL): 1 = no atomics, 2 = no tap loads, 4 = conflict-free LDS addresses (results invalid)
};

// grid (ceil(hw/64), kchunks, Cp/4), block 64: one lane = one pixel x one slice of consecutive depth candidates x
// one 16-byte channel word.  Consecutive candidates of a pixel sample neighbouring positions along its epipolar
// line — for the far planes the SAME 2x2 source cell for several candidates in a row — so the four tap gradients
// are accumulated in registers while the cell stays the same and flushed with atomics only when it changes
// (4-8x fewer atomics than one per (pixel, candidate, view, channel); the kernel is atomic-bound).
__global__ __launch_bounds__(64) void costvol_bwd_kernel(const CostvolBwdArgs a) {
#ifdef NRGBD_DEV
    const int abl = a.abl;
#else
    constexpr int abl = 0;
#endif
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
                    if (acc[tpi][e] != 0.f && !(abl & 1)) atomicAdd(gs + o[tpi] + e, acc[tpi][e]);
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
            float4 A = r, B = r, Cc = r, Dd = r;
            if (!(abl & 2)) {
                A = *reinterpret_cast<const float4*>(sv + o[0]); B = *reinterpret_cast<const float4*>(sv + o[1]);
                Cc = *reinterpret_cast<const float4*>(sv + o[2]); Dd = *reinterpret_cast<const float4*>(sv + o[3]);
            }
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

constexpr int kBwdThreads = 1024;
constexpr size_t kBwdLdsMax = 144 * 1024;   // of the 160 KB of a CU

// grid (kchunks, Cp/4, V), block 1024, LDS 16*h*w bytes: one workgroup = one source view x one 16-byte channel word
// x one slice of consecutive depth candidates, over ALL pixels.  The gradient of that word of the view lives in LDS
// as four component planes [4][h*w] (adjacent pixels sample adjacent texels -> adjacent banks); a lane walks the
// candidates of a pixel, keeps the four tap gradients of the current 2x2 source cell in registers while the cell
// stays the same (the far planes) and adds them into LDS when it changes.
//   part_src [kchunks][V][Cp/4][h*w][4]   partial g_src of this slice (every element written)
//   part_ref [kchunks][V][Cp/4][h*w][4]   partial g_ref of this slice and view
__global__ __launch_bounds__(kBwdThreads) void costvol_bwd_lds_kernel(const CostvolBwdArgs a, float* __restrict__ part_src,
                                                                      float* __restrict__ part_ref) {
#ifdef NRGBD_DEV
    const int abl = a.abl;
#else
    constexpr int abl = 0;
#endif
    extern __shared__ float gl[];
    const int hw = a.h * a.w, words = a.Cp >> 2;
    const int kc = blockIdx.x, i = blockIdx.y, v = blockIdx.z, tid = threadIdx.x;
    for (int t = tid; t < 4 * hw; t += kBwdThreads) gl[t] = 0.f;
    __syncthreads();
    const int per = (a.D + a.kchunks - 1) / a.kchunks;
    const int k_begin = kc * per, k_end = min(a.D, k_begin + per);
    const int ncomp = min(4, a.C - 4 * i);
    const float wf = (float)a.w, hf = (float)a.h;
    const float* sv = a.src + (size_t)v * hw * a.Cp + 4 * i;
    const size_t slab = (((size_t)kc * a.V + v) * words + i) * hw;
    const float inv_sigma_den = a.sigma;

    for (int p = tid; p < hw; p += kBwdThreads) {
        const float rx = a.rays[p], ry = a.rays[hw + p], rz = a.rays[2 * hw + p];
        const float4 r = *reinterpret_cast<const float4*>(a.ref + (size_t)p * a.Cp + 4 * i);
        const SweepTerm st = make_sweep_term(a.KR + 9 * v, a.Kt + 3 * v, rx, ry, rz);
        float gr[4] = {0.f, 0.f, 0.f, 0.f};
        float acc[4][4];
        float cx0 = -1e30f, cy0 = -1e30f;
        int o[4] = {0, 0, 0, 0};
        bool have = false;
        auto flush = [&]() {
            if (!have) return;
#pragma unroll
            for (int tpi = 0; tpi < 4; ++tpi)
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (acc[tpi][e] != 0.f && !(abl & 1)) atomicAdd(gl + e * hw + ((abl & 4) ? min(p + tpi, hw - 1) : o[tpi]), acc[tpi][e]);
        };
        for (int k = k_begin; k < k_end; ++k) {
            const float gk = a.g_cost[(size_t)k * hw + p] / inv_sigma_den;
            float ix, iy;
            sweep_sample_pos(st, a.d_candi[k], a.cx, a.cy, wf, hf, a.align != 0, ix, iy);
            const Bilinear b = bilinear_zeros(ix, iy, a.w, a.h);
            const float x0f = floorf(ix), y0f = floorf(iy);
            if (!(x0f == cx0 && y0f == cy0)) {      // new cell (also taken for NaN positions)
                flush();
                cx0 = x0f; cy0 = y0f; have = true;
                o[0] = b.y0 * a.w + b.x0; o[1] = b.y0 * a.w + b.x1;
                o[2] = b.y1 * a.w + b.x0; o[3] = b.y1 * a.w + b.x1;
#pragma unroll
                for (int tpi = 0; tpi < 4; ++tpi)
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[tpi][e] = 0.f;
            }
            if (gk == 0.f) continue;
            float4 A = r, B = r, Cc = r, Dd = r;
            if (!(abl & 2)) {
                A = *reinterpret_cast<const float4*>(sv + (size_t)o[0] * a.Cp);
                B = *reinterpret_cast<const float4*>(sv + (size_t)o[1] * a.Cp);
                Cc = *reinterpret_cast<const float4*>(sv + (size_t)o[2] * a.Cp);
                Dd = *reinterpret_cast<const float4*>(sv + (size_t)o[3] * a.Cp);
            }
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
        *reinterpret_cast<float4*>(part_ref + (slab + p) * 4) = make_float4(gr[0], gr[1], gr[2], gr[3]);
    }
    __syncthreads();
    for (int t = tid; t < hw; t += kBwdThreads)
        *reinterpret_cast<float4*>(part_src + (slab + t) * 4) = make_float4(gl[t], gl[hw + t], gl[2 * hw + t], gl[3 * hw + t]);
}

// One thread per (view or reference, texel, channel word): adds the slices (and, for the reference image, the views)
// in index order and writes the NHWC gradients.
__global__ __launch_bounds__(256) void costvol_bwd_reduce_kernel(const float* __restrict__ part_src, const float* __restrict__ part_ref,
                                                                 float* __restrict__ g_src, float* __restrict__ g_ref,
                                                                 int V, int words, int hw, int kchunks) {
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t total = (size_t)(V + 1) * hw * words;
    if (idx >= total) return;
    const int word = (int)(idx % words);
    const int t = (int)((idx / words) % hw);
    const int vv = (int)(idx / ((size_t)words * hw));
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    if (vv < V) {
        for (int kc = 0; kc < kchunks; ++kc) {
            const float4 q = *reinterpret_cast<const float4*>(part_src + ((((size_t)kc * V + vv) * words + word) * hw + t) * 4);
            s.x += q.x; s.y += q.y; s.z += q.z; s.w += q.w;
        }
        *reinterpret_cast<float4*>(g_src + ((size_t)vv * hw + t) * (4 * words) + 4 * word) = s;
    } else {
        for (int kc = 0; kc < kchunks; ++kc)
            for (int v = 0; v < V; ++v) {
                const float4 q = *reinterpret_cast<const float4*>(part_ref + ((((size_t)kc * V + v) * words + word) * hw + t) * 4);
                s.x += q.x; s.y += q.y; s.z += q.z; s.w += q.w;
            }
        *reinterpret_cast<float4*>(g_ref + (size_t)t * (4 * words) + 4 * word) = s;
    }
}

// Depth slices of the LDS kernel: as many as keep the grid within one wave of workgroups (one per CU: its LDS
// plane fills most of the CU), at most 16.  0 = the grid does not fit in LDS (global-atomic kernel).
static int bwd_lds_kchunks(int V, int Cp, int D, int h, int w, int* out) {
    *out = 0;
    if ((size_t)h * w * 16 > kBwdLdsMax) return NRGBD_OK;
    int dev = 0, ncu = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e == hipSuccess) e = hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev);
    if (e != hipSuccess) return (int)e;
    int kc = ncu / (V * (Cp >> 2));
    kc = kc < 1 ? 1 : (kc > 16 ? 16 : kc);
    *out = kc > D ? D : kc;
    return NRGBD_OK;
}

}  // namespace nrgbd

extern "C" int nrgbd_costvol_bwd_workspace(int V, int Cp, int D, int h, int w, size_t* bytes) {
    using namespace nrgbd;
    if (!bytes) return NRGBD_E_NULL;
    if (V <= 0 || V > NRGBD_MAX_V || Cp <= 0 || (Cp & 3) || D <= 0 || D > NRGBD_MAX_D || h <= 0 || w <= 0) return NRGBD_E_SHAPE;
    int kc = 0;
    const int rc = bwd_lds_kchunks(V, Cp, D, h, w, &kc);
    if (rc != NRGBD_OK) return rc;
    *bytes = 2 * (size_t)kc * V * (Cp >> 2) * h * w * 4 * sizeof(float);
    return NRGBD_OK;
}

extern "C" int nrgbd_costvol_bwd(const float* ref_nhwc, const float* src_nhwc, const float* KR, const float* Kt,
                                 const float* rays, const float* d_candi, float cx, float cy, float sigma,
                                 int dist, int align_corners, const float* g_cost, float* g_ref, float* g_src,
                                 int V, int C, int Cp, int D, int h, int w, void* workspace, size_t workspace_bytes,
                                 void* stream) {
    using namespace nrgbd;
    if (!ref_nhwc || !src_nhwc || !KR || !Kt || !rays || !d_candi || !g_cost || !g_ref || !g_src) return NRGBD_E_NULL;
    if (V <= 0 || V > NRGBD_MAX_V || C <= 0 || D <= 0 || D > NRGBD_MAX_D || h <= 0 || w <= 0) return NRGBD_E_SHAPE;
    if ((Cp & 3) || Cp < C || Cp - C > 3) return NRGBD_E_ALIGN;
    if (dist != NRGBD_DIST_L2 && dist != NRGBD_DIST_L1) return NRGBD_E_ARG;
    hipStream_t s = (hipStream_t)stream;
    const size_t hw = (size_t)h * w;
    int lds_kc = 0;
    const int rc = bwd_lds_kchunks(V, Cp, D, h, w, &lds_kc);
    if (rc != NRGBD_OK) return rc;
    if (lds_kc > 0) {
        const size_t half = (size_t)lds_kc * V * (Cp >> 2) * hw * 4;   // floats per partial array
        if (!workspace) return NRGBD_E_NULL;
        if (workspace_bytes < 2 * half * sizeof(float)) return NRGBD_E_SHAPE;
        if ((uintptr_t)workspace & 15) return NRGBD_E_ALIGN;
        float* part_src = static_cast<float*>(workspace);
        float* part_ref = part_src + half;
        CostvolBwdArgs a{ref_nhwc, src_nhwc, KR, Kt, rays, d_candi, g_cost, g_ref, g_src, cx, cy, sigma,
                         dist, align_corners, V, C, Cp, D, h, w, lds_kc, dev_env_int("NRGBD_BWD_ABL")};
        const size_t lds = hw * 16;
        hipError_t e = hipSuccess;
        if (lds > 64 * 1024)   // the function's opt-in: always the maximum, never this call's size (a hipGraph replay runs under the current value)
            e = set_max_dynamic_lds(reinterpret_cast<const void*>(&costvol_bwd_lds_kernel), 160 * 1024);
        if (e != hipSuccess) return (int)e;
        hipLaunchKernelGGL(costvol_bwd_lds_kernel, dim3(lds_kc, Cp >> 2, V), dim3(kBwdThreads), lds, s, a, part_src, part_ref);
        NRGBD_CHECK_LAUNCH();
        const size_t total = (size_t)(V + 1) * hw * (Cp >> 2);
        hipLaunchKernelGGL(costvol_bwd_reduce_kernel, dim3((unsigned)ceil_div((long)total, 256)), dim3(256), 0, s,
                           part_src, part_ref, g_src, g_ref, V, Cp >> 2, (int)hw, lds_kc);
        NRGBD_CHECK_LAUNCH();
        return NRGBD_OK;
    }
    hipError_t e = hipMemsetAsync(g_ref, 0, hw * Cp * sizeof(float), s);
    if (e != hipSuccess) return (int)e;
    e = hipMemsetAsync(g_src, 0, (size_t)V * hw * Cp * sizeof(float), s);
    if (e != hipSuccess) return (int)e;
    // slices of consecutive candidates: long enough for the register run-accumulation, parallelism comes from
    // the channel-word axis of the grid
    const int kchunks = D >= 32 ? 4 : (D >= 8 ? 2 : 1);
    CostvolBwdArgs a{ref_nhwc, src_nhwc, KR, Kt, rays, d_candi, g_cost, g_ref, g_src, cx, cy, sigma,
                     dist, align_corners, V, C, Cp, D, h, w, kchunks, dev_env_int("NRGBD_BWD_ABL")};
    hipLaunchKernelGGL(costvol_bwd_kernel, dim3(ceil_div((long)hw, 64), kchunks, Cp >> 2), dim3(64), 0, s, a);
    NRGBD_CHECK_LAUNCH();
    return NRGBD_OK;
}
