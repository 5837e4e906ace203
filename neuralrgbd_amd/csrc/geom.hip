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


// nrgbd_pose_inverse — inverse of the 4x4 camera motion that the PREDICT step resamples through
// (test_utils/test_KVNet.py:50,52: `Src_CamPoses[ibatch, t_win_r].inverse()`).  The reference hands this to the host
// LAPACK (MKL sgetrf + sgetrs on the transposed matrix): an opaque operation order that depends on the library build,
// and a 1-ulp change of the result moves every one of the D*h*w sampling points of the DPV resample (log-probabilities
// with slopes of tens per voxel: 1e-7 of coordinate becomes 1e-2 of BV_predict).  The path therefore owns the operation:
// Gauss-Jordan elimination with partial pivoting on [A | I] in fp64, every operation written out below and rounded once
// (-ffp-contract=off), result rounded to fp32 — i.e. the correctly rounded fp32 inverse up to double rounding, at most
// half an fp32 ulp from exact and therefore at least as close to the reference's own result as that result is to exact.
// oracle/nrgbd_oracle.c::oracle_pose_inverse is the same sequence, operation for operation (bit-identical: IEEE fp64
// add / mul / div on both sides).  One thread per matrix.
__global__ void pose_inverse_kernel(const float* __restrict__ T, long stride, float* __restrict__ out, int* __restrict__ singular,
                                    int n) {
    const int m = blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= n) return;
    double a[4][8];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            a[i][j] = (double)T[m * stride + 4 * i + j];
            a[i][4 + j] = (i == j) ? 1.0 : 0.0;
        }
    bool bad = false;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        int p = c;                                   // partial pivoting: first row of maximal |a[r][c]|, r >= c
        double best = __builtin_fabs(a[c][c]);
#pragma unroll
        for (int r = c + 1; r < 4; ++r) {
            const double v = __builtin_fabs(a[r][c]);
            if (v > best) { best = v; p = r; }
        }
        if (!(best > 0.0)) { bad = true; break; }
#pragma unroll
        for (int r = c + 1; r < 4; ++r)
            if (r == p) {
#pragma unroll
                for (int j = 0; j < 8; ++j) { const double tmp = a[c][j]; a[c][j] = a[r][j]; a[r][j] = tmp; }
            }
        const double piv = a[c][c];
#pragma unroll
        for (int j = 0; j < 8; ++j) a[c][j] = a[c][j] / piv;        // IEEE division, no reciprocal
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (r == c) continue;
            const double f = a[r][c];
#pragma unroll
            for (int j = 0; j < 8; ++j) a[r][j] = a[r][j] - f * a[c][j];   // product and difference rounded separately
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) out[m * 16 + 4 * i + j] = bad ? __builtin_nanf("") : (float)a[i][4 + j];
    if (singular && bad) atomicAdd(singular, 1);
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

extern "C" int nrgbd_pose_inverse(const float* T, long matrix_stride, float* T_inv, int* singular_count, int n, void* stream) {
    if (!T || !T_inv) return NRGBD_E_NULL;
    if (n <= 0 || matrix_stride < 16) return NRGBD_E_SHAPE;
    hipLaunchKernelGGL(nrgbd::pose_inverse_kernel, dim3(nrgbd::ceil_div(n, 64)), dim3(64), 0, (hipStream_t)stream, T,
                       matrix_stride, T_inv, singular_count, n);
    NRGBD_CHECK_LAUNCH();
    return NRGBD_OK;
}
