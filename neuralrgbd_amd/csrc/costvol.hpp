// costvol.hpp — argument block and launchers shared by the cost-volume kernels.
#pragma once
#include "common.hpp"

namespace nrgbd {

struct CostvolArgs {
    const float* ref;      // [h][w][Cp]
    const float* src;      // [V][h][w][Cp]
    const float* KR;       // [V][9]
    const float* Kt;       // [V][3]
    const float* rays;     // [3][hw]
    const float* d_candi;  // [D]
    float* out_cost;       // [D][hw] or null
    float* out_logp;       // [D][hw] or null
    float cx, cy, sigma;
    int dist, align;
    int V, C, Cp, D, h, w;
    int nsingle;           // LDS generation: leading candidates that get a workgroup each (set by the launcher)
    int debug;             // developer bits, honoured only in -DNRGBD_DEV builds (env NRGBD_ABLATE): 1 = no staging, 2 = no math, 4 = XCD-owned tile order, 8 = singles on any grid, (g+1)<<8 = run candidate group g only
    int nchunk, kchunk;    // quad generation: candidate chunks per tile / candidates per chunk (set by the launcher)
    int fuse_softmax;      // quad generation: the workgroup owns all D candidates and also writes out_logp
    float rcx, rcy, rsigma;  // quad generation: RN(1/cx), RN(1/cy), RN(1/sigma) (host, double precision) for div_by_const
    long long* trace;      // developer builds: per-workgroup phase clocks [workgroups][24] (tools/cv_trace.py); null in the product library
};

// Developer ablation bits are compiled out of the product library: a stray environment variable must never change results.
#ifdef NRGBD_DEV
#define NRGBD_DBG(a, bits) ((a).debug & (bits))
#else
#define NRGBD_DBG(a, bits) 0
#endif

enum { NRGBD_GEN_AUTO = 0, NRGBD_GEN_GATHER = 1, NRGBD_GEN_LDS = 2, NRGBD_GEN_QUAD = 3 };

// costvol_lds.hip: LDS-staged generation (returns NRGBD_E_SHAPE when Cp/4 has no instantiation)
int launch_costvol_lds(const CostvolArgs& a, hipStream_t stream);
bool costvol_lds_supported(int cp4);
// costvol_quad.hip: generation 3 (4 lanes per (pixel, candidate), conflict-free LDS taps, fused log-softmax)
bool costvol_quad_supported(const CostvolArgs& a);
int launch_costvol_quad(const CostvolArgs& a, hipStream_t stream, bool* did_softmax);
// softmax.hip
int launch_logsoftmax_d(const float* a, const float* b, float scale, float* out, int D, size_t n,
                        hipStream_t stream);

}  // namespace nrgbd
