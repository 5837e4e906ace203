// warpdepth.hip — photometric warp through a per-pixel depth map (local bundle adjustment), forward and backward.
// Replaces warping/homography.py:479-528 (back_warp_th_Rt_msrc) and :530-575 (back_warp_th_Rt); the backward kernel
// replaces what torch autograd builds for ICP/opt_pose_numerical.py:245-294, where the warped sources are differentiated
// w.r.t. the N rigid motions (R_n, t_n) being refined.  Same sampling core as the plane sweep (zeros padding, ATen
// un-normalisation), with X = dmap[p] ray_p instead of a plane: Y = R X + t, P = K Y, (u, v) = P_xy / P_z (no epsilon,
// homography.py:510), g = (u - cx)/cx.  The matrix products are fma chains over k, the order the reference's 4x4 matmuls
// execute on CPU (oracle/nrgbd_oracle.c::depth_warp_coords, pinned against the live reference).
// HBM-bound: src [N][C][H][W] is gathered (L2-resident, 4 taps x C), out written once, coalesced along x.
#include "common.hpp"

namespace nrgbd {

struct DepthWarpArgs {
    const float* src; const float* dmap; const float* K; const float* R; const float* t; const float* rays;
    const float* g_out; float* out; float* partial;
    int N, C, H, W, nwg;
};

struct DepthWarpPoint { float X[3], P[3], ix, iy; };

__device__ __forceinline__ DepthWarpPoint depth_warp_point(const float* __restrict__ K, const float* __restrict__ R,
                                                           const float* __restrict__ t, float rx, float ry, float rz,
                                                           float d, int W, int H) {
    DepthWarpPoint q;
    q.X[0] = d * rx; q.X[1] = d * ry; q.X[2] = d * rz;
    float Y[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        float s = R[3 * i] * q.X[0];
        s = __builtin_fmaf(R[3 * i + 1], q.X[1], s);
        s = __builtin_fmaf(R[3 * i + 2], q.X[2], s);
        Y[i] = __builtin_fmaf(t[i], 1.0f, s);
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        float s = K[3 * i] * Y[0];
        s = __builtin_fmaf(K[3 * i + 1], Y[1], s);
        s = __builtin_fmaf(K[3 * i + 2], Y[2], s);
        q.P[i] = s;
    }
    const float u = q.P[0] / q.P[2], v = q.P[1] / q.P[2];
    const float cx = K[2], cy = K[5];
    q.ix = unnormalize((u - cx) / cx, (float)W, false);
    q.iy = unnormalize((v - cy) / cy, (float)H, false);
    return q;
}

// grid (ceil(HW/256), N): one lane = one reference pixel of one view
__global__ __launch_bounds__(256) void warp_depth_fwd_kernel(const DepthWarpArgs a) {
    const size_t hw = (size_t)a.H * a.W;
    const size_t p = (size_t)blockIdx.x * 256 + threadIdx.x;
    const int n = blockIdx.y;
    if (p >= hw) return;
    const DepthWarpPoint q = depth_warp_point(a.K, a.R + 9 * n, a.t + 3 * n, a.rays[p], a.rays[hw + p], a.rays[2 * hw + p],
                                              a.dmap[p], a.W, a.H);
    const Bilinear b = bilinear_zeros(q.ix, q.iy, a.W, a.H);
    const size_t onw = (size_t)b.y0 * a.W + b.x0, one = (size_t)b.y0 * a.W + b.x1;
    const size_t osw = (size_t)b.y1 * a.W + b.x0, ose = (size_t)b.y1 * a.W + b.x1;
    for (int c = 0; c < a.C; ++c) {
        const float* pl = a.src + ((size_t)n * a.C + c) * hw;
        a.out[((size_t)n * a.C + c) * hw + p] = lerp4(pl[onw], pl[one], pl[osw], pl[ose], b);
    }
}

// d sum(out g_out) / d(R_n, t_n): per lane the 12 terms dY (x) [X 1], reduced per workgroup into partial[n][wg][12]
__global__ __launch_bounds__(256) void warp_depth_bwd_kernel(const DepthWarpArgs a) {
    __shared__ float red[4][12];
    const size_t hw = (size_t)a.H * a.W;
    const size_t p = (size_t)blockIdx.x * 256 + threadIdx.x;
    const int n = blockIdx.y;
    float g12[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) g12[i] = 0.f;
    if (p < hw) {
        const DepthWarpPoint q = depth_warp_point(a.K, a.R + 9 * n, a.t + 3 * n, a.rays[p], a.rays[hw + p],
                                                  a.rays[2 * hw + p], a.dmap[p], a.W, a.H);
        const Bilinear b = bilinear_zeros(q.ix, q.iy, a.W, a.H);   // weights already zero for taps outside the image
        const float x0f = floorf(q.ix), y0f = floorf(q.iy);
        const float fx = q.ix - x0f, fy = q.iy - y0f;
        // in-bounds masks of the four taps (ATen grid_sampler_2d_backward adds a tap's term only when it is inside)
        const float wm = (float)(a.W - 1), hm = (float)(a.H - 1);
        const bool vx0 = (x0f >= 0.f) && (x0f <= wm), vx1 = (x0f + 1.f >= 0.f) && (x0f + 1.f <= wm);
        const bool vy0 = (y0f >= 0.f) && (y0f <= hm), vy1 = (y0f + 1.f >= 0.f) && (y0f + 1.f <= hm);
        const size_t onw = (size_t)b.y0 * a.W + b.x0, one = (size_t)b.y0 * a.W + b.x1;
        const size_t osw = (size_t)b.y1 * a.W + b.x0, ose = (size_t)b.y1 * a.W + b.x1;
        float gix = 0.f, giy = 0.f;
        for (int c = 0; c < a.C; ++c) {
            const float* pl = a.src + ((size_t)n * a.C + c) * hw;
            const float g = a.g_out[((size_t)n * a.C + c) * hw + p];
            const float v00 = (vx0 && vy0) ? pl[onw] : 0.f, v01 = (vx1 && vy0) ? pl[one] : 0.f;
            const float v10 = (vx0 && vy1) ? pl[osw] : 0.f, v11 = (vx1 && vy1) ? pl[ose] : 0.f;
            gix = __builtin_fmaf(g, __builtin_fmaf(v11 - v10, fy, (v01 - v00) * (1.f - fy)), gix);
            giy = __builtin_fmaf(g, __builtin_fmaf(v11 - v01, fx, (v10 - v00) * (1.f - fx)), giy);
        }
        const float cx = a.K[2], cy = a.K[5];
        const float du = gix * (0.5f * (float)a.W) / cx, dv = giy * (0.5f * (float)a.H) / cy;
        const float ipz = 1.f / q.P[2];
        const float dP0 = du * ipz, dP1 = dv * ipz, dP2 = -(du * q.P[0] + dv * q.P[1]) * ipz * ipz;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const float dY = a.K[i] * dP0 + a.K[3 + i] * dP1 + a.K[6 + i] * dP2;   // (K^T dP)_i
            g12[3 * i + 0] = dY * q.X[0]; g12[3 * i + 1] = dY * q.X[1]; g12[3 * i + 2] = dY * q.X[2];
            g12[9 + i] = dY;
        }
    }
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < 12; ++i) {
        const float s = wave_sum(g12[i]);
        if (lane == 0) red[wv][i] = s;
    }
    __syncthreads();
    if (threadIdx.x < 12)
        a.partial[((size_t)n * a.nwg + blockIdx.x) * 12 + threadIdx.x] =
            (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}

// partial[n][nwg][12] -> g_R[n][9], g_t[n][3]: fixed-order double-precision sum (bitwise reproducible)
__global__ __launch_bounds__(64) void warp_depth_bwd_finalize(const float* __restrict__ partial, int nwg,
                                                              float* __restrict__ g_R, float* __restrict__ g_t) {
    const int n = blockIdx.x, i = threadIdx.x;
    if (i >= 12) return;
    double s = 0.0;
    for (int w = 0; w < nwg; ++w) s += (double)partial[((size_t)n * nwg + w) * 12 + i];
    if (i < 9) g_R[9 * n + i] = (float)s;
    else g_t[3 * n + (i - 9)] = (float)s;
}

}  // namespace nrgbd

extern "C" int nrgbd_warp_depth_fwd(const float* src, const float* dmap, const float* K, const float* R, const float* t,
                                    const float* rays, float* out, int N, int C, int H, int W, void* stream) {
    using namespace nrgbd;
    if (!src || !dmap || !K || !R || !t || !rays || !out) return NRGBD_E_NULL;
    if (N <= 0 || N > 65535 || C <= 0 || H <= 0 || W <= 0) return NRGBD_E_SHAPE;
    DepthWarpArgs a{src, dmap, K, R, t, rays, nullptr, out, nullptr, N, C, H, W, 0};
    hipLaunchKernelGGL(warp_depth_fwd_kernel, dim3(ceil_div((long)H * W, 256), N), dim3(256), 0, (hipStream_t)stream, a);
    NRGBD_CHECK_LAUNCH();
    return NRGBD_OK;
}

extern "C" int nrgbd_warp_depth_bwd_workgroups(int H, int W) { return nrgbd::ceil_div((long)H * W, 256); }

extern "C" int nrgbd_warp_depth_bwd(const float* src, const float* dmap, const float* K, const float* R, const float* t,
                                    const float* rays, const float* g_out, float* partial, float* g_R, float* g_t, int N,
                                    int C, int H, int W, void* stream) {
    using namespace nrgbd;
    if (!src || !dmap || !K || !R || !t || !rays || !g_out || !partial || !g_R || !g_t) return NRGBD_E_NULL;
    if (N <= 0 || N > 65535 || C <= 0 || H <= 0 || W <= 0) return NRGBD_E_SHAPE;
    const int nwg = ceil_div((long)H * W, 256);
    DepthWarpArgs a{src, dmap, K, R, t, rays, g_out, nullptr, partial, N, C, H, W, nwg};
    hipLaunchKernelGGL(warp_depth_bwd_kernel, dim3(nwg, N), dim3(256), 0, (hipStream_t)stream, a);
    NRGBD_CHECK_LAUNCH();
    hipLaunchKernelGGL(warp_depth_bwd_finalize, dim3(N), dim3(64), 0, (hipStream_t)stream, partial, nwg, g_R, g_t);
    NRGBD_CHECK_LAUNCH();
    return NRGBD_OK;
}
