// conv3d_c1_bwd.hip — both gradients of the K-Net's last layer Conv3d(64, 1, 3, padding 1) (models/basic.py:92-94) for the training path.
//
// Why.  Until round 6 the training path ran this layer zero-padded to 64 outputs so that forward, data gradient and weight gradient could use
// the 64 -> 64 matrix-core kernels: three launches of a K-Net layer (0.23 + 0.23 + 0.57 ms at the training grid) of which 63/64 is
// multiplication by zero.  With ONE output channel all three directions are a 27-tap stencil per voxel — 1,728 multiply-adds, i.e. memory
// bound: the forward is csrc/conv3d.hip's depth-marching kernel (nrgbd_conv3d_3x3x3_cout1_f32); here are the other two.
//
//   y[v]        = sum_tap sum_ci x[v + off(tap)][ci] w[ci][tap]                      off(tap) = (kd - 1, kh - 1, kw - 1)
//   gx[v][ci]   = sum_tap gy[v - off(tap)] w[ci][tap]                                (dgrad: one 16-byte store per thread)
//   gw[ci][tap] = sum_v   gy[v - off(tap)] x[v][ci]                                  (wgrad: 27 x 4 running sums per thread)
//
// Both walk 8 x 16-voxel tiles of one depth slice: the 3 x 10 x 18 halo of gy (one channel) sits in LDS, a thread = (voxel column of 16,
// 4 consecutive channels) reads its 27 neighbours as LDS broadcasts and touches x / gx once, as whole 16-byte words of a channels-last voxel.
// Summation orders are fixed (taps ascending; tiles in list order; workgroup partials in index order): bitwise reproducible.
#include "common.hpp"

namespace nrgbd {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kB1TH = 8, kB1TW = 16;            // voxels of one slice per tile
constexpr int kB1GP = 20;                       // floats per halo row in LDS (18 + 2 pad)
constexpr int kB1G = 3 * (kB1TH + 2) * kB1GP;   // 600 floats

__device__ __forceinline__ void c1_tile(int t, int tilesH, int tilesW, int& d, int& h0, int& w0) {
    const int tw = t % tilesW; t /= tilesW;
    const int th = t % tilesH;
    d = t / tilesH; h0 = th * kB1TH; w0 = tw * kB1TW;
}

__device__ __forceinline__ void c1_load_gy(float* g, const float* __restrict__ gy, int d, int h0, int w0, int D, int H, int W, int tid) {
    for (int i = tid; i < 3 * (kB1TH + 2) * (kB1TW + 2); i += 256) {
        const int dz = i / ((kB1TH + 2) * (kB1TW + 2)), rem = i - dz * ((kB1TH + 2) * (kB1TW + 2));
        const int r = rem / (kB1TW + 2), c = rem - r * (kB1TW + 2);
        const int zz = d + dz - 1, yy = h0 + r - 1, xx = w0 + c - 1;
        const bool in = zz >= 0 && zz < D && yy >= 0 && yy < H && xx >= 0 && xx < W;
        g[(dz * (kB1TH + 2) + r) * kB1GP + c] = in ? gy[((size_t)zz * H + yy) * W + xx] : 0.f;
    }
}

// neighbour gy[v - off(tap)] of the voxel (row r, column vx) of the tile: slice 2 - kd, row r + 2 - kh, column vx + 2 - kw of the halo
__device__ __forceinline__ float c1_nb(const float* g, int r, int vx, int kd, int kh, int kw) {
    return g[((2 - kd) * (kB1TH + 2) + r + 2 - kh) * kB1GP + vx + 2 - kw];
}

// gx [D][H][W][64] = data gradient; w_tm [27][64] tap-major weights (w[0][ci][tap] transposed)
__global__ __launch_bounds__(256) void conv3d_cout1_dgrad_kernel(const float* __restrict__ gy, const float* __restrict__ w_tm,
                                                                 float* __restrict__ gx, int D, int H, int W) {
    __shared__ float g[kB1G];
    const int tid = threadIdx.x, cq = tid & 15, vx = tid >> 4;
    f32x4 wv[27];
#pragma unroll
    for (int t = 0; t < 27; ++t) wv[t] = *reinterpret_cast<const f32x4*>(w_tm + t * 64 + cq * 4);
    const int tilesW = (W + kB1TW - 1) / kB1TW, tilesH = (H + kB1TH - 1) / kB1TH, ntiles = D * tilesH * tilesW;
    for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
        int d, h0, w0;
        c1_tile(t, tilesH, tilesW, d, h0, w0);
        __syncthreads();                                   // the previous tile's readers are done
        c1_load_gy(g, gy, d, h0, w0, D, H, W, tid);
        __syncthreads();
        const int w = w0 + vx;
#pragma unroll 2
        for (int r = 0; r < kB1TH; ++r) {
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kd = 0; kd < 3; ++kd)
#pragma unroll
                for (int kh = 0; kh < 3; ++kh)
#pragma unroll
                    for (int kw = 0; kw < 3; ++kw) {
                        const float n = c1_nb(g, r, vx, kd, kh, kw);
                        const f32x4 q = wv[(kd * 3 + kh) * 3 + kw];
                        acc.x = __builtin_fmaf(n, q.x, acc.x); acc.y = __builtin_fmaf(n, q.y, acc.y);
                        acc.z = __builtin_fmaf(n, q.z, acc.z); acc.w = __builtin_fmaf(n, q.w, acc.w);
                    }
            const int h = h0 + r;
            if (h < H && w < W) *reinterpret_cast<f32x4*>(gx + (((size_t)d * H + h) * W + w) * 64 + cq * 4) = acc;
        }
    }
}

// partial [gridDim.x][27][64]: this workgroup's sums over its tiles
__global__ __launch_bounds__(256) void conv3d_cout1_wgrad_kernel(const float* __restrict__ x, const float* __restrict__ gy,
                                                                 float* __restrict__ partial, int D, int H, int W) {
    __shared__ float g[kB1G];
    __shared__ f32x4 red[4][27][16];
    const int tid = threadIdx.x, cq = tid & 15, vx = tid >> 4, lane = tid & 63, wave = tid >> 6;
    f32x4 acc[27];
#pragma unroll
    for (int t = 0; t < 27; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int tilesW = (W + kB1TW - 1) / kB1TW, tilesH = (H + kB1TH - 1) / kB1TH, ntiles = D * tilesH * tilesW;
    for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
        int d, h0, w0;
        c1_tile(t, tilesH, tilesW, d, h0, w0);
        __syncthreads();
        c1_load_gy(g, gy, d, h0, w0, D, H, W, tid);
        __syncthreads();
        const int w = w0 + vx;
#pragma unroll 2
        for (int r = 0; r < kB1TH; ++r) {
            const int h = h0 + r;
            f32x4 xv = {0.f, 0.f, 0.f, 0.f};
            if (h < H && w < W) xv = *reinterpret_cast<const f32x4*>(x + (((size_t)d * H + h) * W + w) * 64 + cq * 4);
#pragma unroll
            for (int kd = 0; kd < 3; ++kd)
#pragma unroll
                for (int kh = 0; kh < 3; ++kh)
#pragma unroll
                    for (int kw = 0; kw < 3; ++kw) {
                        const float n = c1_nb(g, r, vx, kd, kh, kw);
                        f32x4& a = acc[(kd * 3 + kh) * 3 + kw];
                        a.x = __builtin_fmaf(n, xv.x, a.x); a.y = __builtin_fmaf(n, xv.y, a.y);
                        a.z = __builtin_fmaf(n, xv.z, a.z); a.w = __builtin_fmaf(n, xv.w, a.w);
                    }
        }
    }
    // the 16 voxel columns of a channel quad: 4 inside a wave (lane bits 4, 5), 4 waves through LDS, both in fixed order
#pragma unroll
    for (int t = 0; t < 27; ++t) {
        f32x4 a = acc[t];
#pragma unroll
        for (int m = 16; m <= 32; m <<= 1) {
            a.x += __shfl_xor(a.x, m, 64); a.y += __shfl_xor(a.y, m, 64);
            a.z += __shfl_xor(a.z, m, 64); a.w += __shfl_xor(a.w, m, 64);
        }
        if (lane < 16) red[wave][t][cq] = a;
    }
    __syncthreads();
    for (int i = tid; i < 27 * 16; i += 256) {
        const int t = i >> 4, c = i & 15;
        const f32x4 s = ((red[0][t][c] + red[1][t][c]) + red[2][t][c]) + red[3][t][c];
        *reinterpret_cast<f32x4*>(partial + (size_t)blockIdx.x * (27 * 64) + t * 64 + c * 4) = s;
    }
}

// dw [64][27] (= torch's [1][64][3][3][3]) = sum over the workgroup partials: 64 outputs x 4 interleaved slices per workgroup, combined in order
__global__ __launch_bounds__(256) void conv3d_cout1_wgrad_reduce_kernel(const float* __restrict__ partial, float* __restrict__ dw, int nwg) {
    __shared__ float part[4][64];
    const int o = threadIdx.x & 63, sl = threadIdx.x >> 6;
    const int i = blockIdx.x * 64 + o;                 // index (tap, ci) of the partial layout
    float s = 0.f;
    for (int gI = sl; gI < nwg; gI += 4) s += partial[(size_t)gI * (27 * 64) + i];
    part[sl][o] = s;
    __syncthreads();
    if (sl == 0) {
        const int tap = i >> 6, ci = i & 63;
        dw[ci * 27 + tap] = ((part[0][o] + part[1][o]) + part[2][o]) + part[3][o];
    }
}

static int c1_bwd_workgroups(int D, int H, int W) {
    const long nt = (long)D * ceil_div(H, kB1TH) * ceil_div(W, kB1TW);
    return (int)(nt < 512 ? nt : 512);
}

}  // namespace nrgbd

extern "C" int nrgbd_conv3d_cout1_dgrad_f32(const float* gy, const float* w_tap_major, float* gx, int D, int H, int W, void* stream) {
    using namespace nrgbd;
    if (!gy || !w_tap_major || !gx) return NRGBD_E_NULL;
    if (D <= 0 || H <= 0 || W <= 0 || (long)D * H * W >= (1L << 31)) return NRGBD_E_SHAPE;
    if ((reinterpret_cast<uintptr_t>(gx) | reinterpret_cast<uintptr_t>(w_tap_major)) & 15) return NRGBD_E_ALIGN;
    const long nt = (long)D * ceil_div(H, kB1TH) * ceil_div(W, kB1TW);
    hipLaunchKernelGGL(conv3d_cout1_dgrad_kernel, dim3((unsigned)(nt < 2048 ? nt : 2048)), dim3(256), 0, (hipStream_t)stream, gy, w_tap_major,
                       gx, D, H, W);
    NRGBD_CHECK_LAUNCH();
    return NRGBD_OK;
}

extern "C" int nrgbd_conv3d_cout1_wgrad_workspace(int D, int H, int W, size_t* bytes) {
    using namespace nrgbd;
    if (!bytes) return NRGBD_E_NULL;
    if (D <= 0 || H <= 0 || W <= 0 || (long)D * H * W >= (1L << 31)) return NRGBD_E_SHAPE;
    *bytes = (size_t)c1_bwd_workgroups(D, H, W) * 27 * 64 * sizeof(float);
    return NRGBD_OK;
}

extern "C" int nrgbd_conv3d_cout1_wgrad_f32(const float* x, const float* gy, float* dw, void* workspace, size_t workspace_bytes, int D,
                                            int H, int W, void* stream) {
    using namespace nrgbd;
    if (!x || !gy || !dw || !workspace) return NRGBD_E_NULL;
    if (D <= 0 || H <= 0 || W <= 0 || (long)D * H * W >= (1L << 31)) return NRGBD_E_SHAPE;
    if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(workspace)) & 15) return NRGBD_E_ALIGN;
    const int nwg = c1_bwd_workgroups(D, H, W);
    if (workspace_bytes < (size_t)nwg * 27 * 64 * sizeof(float)) return NRGBD_E_NULL;
    float* partial = static_cast<float*>(workspace);
    hipLaunchKernelGGL(conv3d_cout1_wgrad_kernel, dim3(nwg), dim3(256), 0, (hipStream_t)stream, x, gy, partial, D, H, W);
    NRGBD_CHECK_LAUNCH();
    hipLaunchKernelGGL(conv3d_cout1_wgrad_reduce_kernel, dim3(27), dim3(256), 0, (hipStream_t)stream, partial, dw, nwg);
    NRGBD_CHECK_LAUNCH();
    return NRGBD_OK;
}
