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
    int debug;             // developer bits (env NRGBD_ABLATE): 1 = no staging, 2 = no math, 4 = XCD-owned tile order, 8 = singles on any grid, (g+1)<<8 = run candidate group g only
};

// costvol_lds.hip: LDS-staged generation (returns NRGBD_E_SHAPE when Cp/4 has no instantiation)
int launch_costvol_lds(const CostvolArgs& a, hipStream_t stream);
bool costvol_lds_supported(int cp4);
// softmax.hip
int launch_logsoftmax_d(const float* a, const float* b, float scale, float* out, int D, size_t n,
                        hipStream_t stream);

}  // namespace nrgbd
