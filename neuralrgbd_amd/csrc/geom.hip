// geom.hip — per-window homography terms on the device.
// Replaces the two tiny matmuls of warping/homography.py:315-317 (term1 = K t_v, left factor K R_v of term2), which the
// host mirror used to hand to rocBLAS: a vendor GEMM is free to pick its own summation order for a K=3 contraction, and a
// 1-ulp change of K R_v moves every sampling coordinate of the view.  Here the order is the one the reference's CPU path
// executes under torch 2.10 (probed, tests/test_host.py::test_homography_terms_order): K R_v is an fma chain over k = 0,1,2
// (sgemm micro-kernel), K t_v is ((K[i][1] t[1] + K[i][2] t[2]) + K[i][0] t[0]) with separately rounded products (the
// 3-element sgemv path), so KR / Kt — and with them tap selection — agree with the golden vectors bit for bit.
#include "common.hpp"

namespace nrgbd {

__global__ void homography_terms_kernel(const float* __restrict__ K, const float* __restrict__ R, long rs_v, long rs_r,
                                        const float* __restrict__ t, long ts_v, long ts_e, float* __restrict__ KR,
                                        float* __restrict__ Kt, int V) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 12 * V) return;
    const int v = i / 12, e = i - 12 * v;
    if (e < 9) {
        const int r = e / 3, c = e - 3 * r;
        const float* Rv = R + v * rs_v;
        float s = K[3 * r] * Rv[c];
        s = __builtin_fmaf(K[3 * r + 1], Rv[rs_r + c], s);
        s = __builtin_fmaf(K[3 * r + 2], Rv[2 * rs_r + c], s);
        KR[9 * v + e] = s;
    } else {
        const int r = e - 9;
        const float* tv = t + v * ts_v;
        const float p0 = K[3 * r] * tv[0], p1 = K[3 * r + 1] * tv[ts_e], p2 = K[3 * r + 2] * tv[2 * ts_e];
        Kt[3 * v + r] = (p1 + p2) + p0;
    }
}

}  // namespace nrgbd

extern "C" int nrgbd_homography_terms(const float* K, const float* R, long r_view_stride, long r_row_stride,
                                      const float* t, long t_view_stride, long t_elem_stride, float* KR, float* Kt,
                                      int V, void* stream) {
    if (!K || !R || !t || !KR || !Kt) return NRGBD_E_NULL;
    if (V <= 0 || V > NRGBD_MAX_V) return NRGBD_E_SHAPE;
    hipLaunchKernelGGL(nrgbd::homography_terms_kernel, dim3(nrgbd::ceil_div(12 * V, 64)), dim3(64), 0,
                       (hipStream_t)stream, K, R, r_view_stride, r_row_stride, t, t_view_stride, t_elem_stride, KR, Kt, V);
    NRGBD_CHECK_LAUNCH();
    return NRGBD_OK;
}
