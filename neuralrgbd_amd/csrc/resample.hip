// resample.hip — PREDICT: rigid 3-D resample of the depth-probability volume (K10).
// Replaces warping/homography.py:654-723 (resample_vol_cuda, d_candi_new = None or given), :873-887
// (_set_vol_border) and the clamp of test_utils/test_KVNet.py:54-59.
//
// The reference builds the [1,D,h,w,3] back-projected point grid on the HOST in a Python loop and
// copies it to the device every frame (:673-682), then syncs twice for z min/max (:689-690).
// Here the point of voxel (k,y,x) is generated in-kernel from the ray table and d_candi, the pose
// is read from device memory, and the border overwrite is applied on the fly to the 8 taps, so the
// step is one HBM-bound launch (read D*hw, write D*hw floats) with no host round trip.
#include "common.hpp"

namespace nrgbd {

struct ResampleArgs {
    const float* dpv; const float* T; const float* rays; const float* d_candi;
    float* out;
    float tan_hh, tan_hv, z_half, z_radius, pad, lo, hi;
    int do_clamp, D, h, w;   // D = planes of the source volume; the launch's grid.y = planes of the output (d_candi's length)
};

// ATen GridSampler.h clip_coordinates: min(size-1, max(x, 0)) with std::min/max NaN behaviour
__device__ __forceinline__ float clip_border(float x, float hi) {
    x = (x < 0.f) ? 0.f : x;
    return (x < hi) ? x : hi;
}

__global__ __launch_bounds__(256) void dpv_resample_kernel(const ResampleArgs a) {
    const size_t hw = (size_t)a.h * a.w;
    const size_t p = (size_t)blockIdx.x * 256 + threadIdx.x;
    const int k = blockIdx.y;
    if (p >= hw) return;
    const float d = a.d_candi[k];
    // homography.py:679-682  X = d * ray
    const float X = d * a.rays[p], Y = d * a.rays[hw + p], Z = d * a.rays[2 * hw + p];
    // :698-702  rel_extM @ [X Y Z 1]^T  (K=4 fma chain)
    float q[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        float acc = a.T[4 * r] * X;
        acc = __builtin_fmaf(a.T[4 * r + 1], Y, acc);
        acc = __builtin_fmaf(a.T[4 * r + 2], Z, acc);
        acc = __builtin_fmaf(a.T[4 * r + 3], 1.f, acc);
        q[r] = acc;
    }
    // :705-710
    const float den = q[2] + 1e-10f;
    const float wq = q[3] + 1e-10f;
    const float gx = ((q[0] / den) / a.tan_hh) / wq;
    const float gy = ((q[1] / den) / a.tan_hv) / wq;
    const float gz = ((q[2] - a.z_half) / a.z_radius) / wq;
    // :716  F.grid_sample 3-D: bilinear, padding 'border', align_corners=False
    const float fx = clip_border(unnormalize(gx, (float)a.w, false), (float)(a.w - 1));
    const float fy = clip_border(unnormalize(gy, (float)a.h, false), (float)(a.h - 1));
    const float fz = clip_border(unnormalize(gz, (float)a.D, false), (float)(a.D - 1));
    const float x0f = floorf(fx), y0f = floorf(fy), z0f = floorf(fz);
    const int x0 = (int)x0f, y0 = (int)y0f, z0 = (int)z0f;
    const float ex = (x0f + 1.f) - fx, ey = (y0f + 1.f) - fy, ez = (z0f + 1.f) - fz;
    const float wx = fx - x0f, wy = fy - y0f, wz = fz - z0f;

    // a tap on one of the 6 faces reads pad (:873-887); a tap beyond size-1 is dropped
    auto tap = [&](int zz, int yy, int xx) -> float {
        const bool face = (zz == 0) | (yy == 0) | (xx == 0) | (zz == a.D - 1) | (yy == a.h - 1) | (xx == a.w - 1);
        return face ? a.pad : a.dpv[((size_t)zz * a.h + yy) * a.w + xx];
    };
    const bool vx1 = x0 + 1 < a.w, vy1 = y0 + 1 < a.h, vz1 = z0 + 1 < a.D;
    const int x1 = vx1 ? x0 + 1 : x0, y1 = vy1 ? y0 + 1 : y0, z1 = vz1 ? z0 + 1 : z0;
    float acc = 0.f;
    acc += tap(z0, y0, x0) * (ex * ey * ez);
    if (vx1) acc += tap(z0, y0, x1) * (wx * ey * ez);
    if (vy1) acc += tap(z0, y1, x0) * (ex * wy * ez);
    if (vx1 && vy1) acc += tap(z0, y1, x1) * (wx * wy * ez);
    if (vz1) acc += tap(z1, y0, x0) * (ex * ey * wz);
    if (vz1 && vx1) acc += tap(z1, y0, x1) * (wx * ey * wz);
    if (vz1 && vy1) acc += tap(z1, y1, x0) * (ex * wy * wz);
    if (vz1 && vx1 && vy1) acc += tap(z1, y1, x1) * (wx * wy * wz);
    if (a.do_clamp) acc = fminf(fmaxf(acc, a.lo), a.hi);
    a.out[(size_t)k * hw + p] = acc;
}

}  // namespace nrgbd

extern "C" int nrgbd_dpv_resample_to(const float* dpv, const float* T, const float* rays,
                                     const float* d_candi_out, float tan_hh, float tan_hv, float z_half,
                                     float z_radius, float pad_value, int do_clamp, float clamp_lo,
                                     float clamp_hi, float* out, int D_src, int D_out, int h, int w, void* stream) {
    using namespace nrgbd;
    if (!dpv || !T || !rays || !d_candi_out || !out) return NRGBD_E_NULL;
    if (dpv == out) return NRGBD_E_ARG;
    if (D_src <= 0 || D_src > 65535 || D_out <= 0 || D_out > 65535 || h <= 0 || w <= 0) return NRGBD_E_SHAPE;
    ResampleArgs a{dpv, T, rays, d_candi_out, out, tan_hh, tan_hv, z_half, z_radius, pad_value,
                   clamp_lo, clamp_hi, do_clamp, D_src, h, w};
    dim3 grid(ceil_div((long)h * w, 256), D_out);
    hipLaunchKernelGGL(dpv_resample_kernel, grid, dim3(256), 0, (hipStream_t)stream, a);
    NRGBD_CHECK_LAUNCH();
    return NRGBD_OK;
}

extern "C" int nrgbd_dpv_resample(const float* dpv, const float* T, const float* rays,
                                  const float* d_candi, float tan_hh, float tan_hv, float z_half,
                                  float z_radius, float pad_value, int do_clamp, float clamp_lo,
                                  float clamp_hi, float* out, int D, int h, int w, void* stream) {
    return nrgbd_dpv_resample_to(dpv, T, rays, d_candi, tan_hh, tan_hv, z_half, z_radius, pad_value, do_clamp, clamp_lo,
                                 clamp_hi, out, D, D, h, w, stream);
}
