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

// grid (ceil(hw/64), kchunks), block 64: one lane = one pixel x one slice of the depth candidates
__global__ __launch_bounds__(64) void costvol_bwd_kernel(const CostvolBwdArgs a) {
    const size_t hw = (size_t)a.h * a.w;
    const size_t p = (size_t)blockIdx.x * 64 + threadIdx.x;
    if (p >= hw) return;
    const int per = (a.D + a.kchunks - 1) / a.kchunks;
    const int k_begin = blockIdx.y * per, k_end = min(a.D, k_begin + per);
    const int cp4 = a.Cp >> 2;
    const float rx = a.rays[p], ry = a.rays[hw + p], rz = a.rays[2 * hw + p];
    const float wf = (float)a.w, hf = (float)a.h;
    const float4* refp = reinterpret_cast<const float4*>(a.ref + p * a.Cp);
    for (int i = 0; i < cp4; ++i) {
        const float4 r = refp[i];
        float4 gr = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int k = k_begin; k < k_end; ++k) {
            const float gk = a.g_cost[(size_t)k * hw + p] / a.sigma;
            if (gk == 0.f) continue;
            const float d = a.d_candi[k];
            for (int v = 0; v < a.V; ++v) {
                const SweepTerm st = make_sweep_term(a.KR + 9 * v, a.Kt + 3 * v, rx, ry, rz);
                float ix, iy;
                sweep_sample_pos(st, d, a.cx, a.cy, wf, hf, a.align != 0, ix, iy);
                const Bilinear b = bilinear_zeros(ix, iy, a.w, a.h);
                const size_t onw = ((size_t)b.y0 * a.w + b.x0) * a.Cp + 4 * i, one = ((size_t)b.y0 * a.w + b.x1) * a.Cp + 4 * i;
                const size_t osw = ((size_t)b.y1 * a.w + b.x0) * a.Cp + 4 * i, ose = ((size_t)b.y1 * a.w + b.x1) * a.Cp + 4 * i;
                const float* sv = a.src + (size_t)v * hw * a.Cp;
                float* gs = a.g_src + (size_t)v * hw * a.Cp;
                const float4 A = *reinterpret_cast<const float4*>(sv + onw), B = *reinterpret_cast<const float4*>(sv + one);
                const float4 Cc = *reinterpret_cast<const float4*>(sv + osw), Dd = *reinterpret_cast<const float4*>(sv + ose);
                const float df[4] = {lerp4(A.x, B.x, Cc.x, Dd.x, b) - r.x, lerp4(A.y, B.y, Cc.y, Dd.y, b) - r.y,
                                     lerp4(A.z, B.z, Cc.z, Dd.z, b) - r.z, lerp4(A.w, B.w, Cc.w, Dd.w, b) - r.w};
                float* grp = &gr.x;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if (4 * i + e >= a.C) continue;
                    const float ds = (a.dist == NRGBD_DIST_L2) ? 2.f * df[e] : (df[e] > 0.f ? 1.f : (df[e] < 0.f ? -1.f : 0.f));
                    const float c = ds * gk;
                    grp[e] -= c;
                    if (b.nw != 0.f) atomicAdd(gs + onw + e, b.nw * c);
                    if (b.ne != 0.f) atomicAdd(gs + one + e, b.ne * c);
                    if (b.sw != 0.f) atomicAdd(gs + osw + e, b.sw * c);
                    if (b.se != 0.f) atomicAdd(gs + ose + e, b.se * c);
                }
            }
        }
        float* go = a.g_ref + p * a.Cp + 4 * i;
        atomicAdd(go + 0, gr.x); atomicAdd(go + 1, gr.y); atomicAdd(go + 2, gr.z); atomicAdd(go + 3, gr.w);
    }
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
    const int kchunks = D < 16 ? D : 16;
    CostvolBwdArgs a{ref_nhwc, src_nhwc, KR, Kt, rays, d_candi, g_cost, g_ref, g_src, cx, cy, sigma,
                     dist, align_corners, V, C, Cp, D, h, w, kchunks};
    hipLaunchKernelGGL(costvol_bwd_kernel, dim3(ceil_div((long)hw, 64), kchunks), dim3(64), 0, s, a);
    NRGBD_CHECK_LAUNCH();
    return NRGBD_OK;
}
