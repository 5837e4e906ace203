// conv3d_wgrad.hip — weight gradient of the K-Net 3x3x3 convolution on the fp32 matrix cores (training).
//
//   dW[co][ci][tap] = sum_voxels  dY[v][co] * X[v + tap][ci]          (X zero outside the volume)
// i.e. 27 skinny GEMMs (64 x Cin, K = D*H*W voxels) that share their operands.  The vendor path (im2col + GEMM /
// CK batched bwd-weight) needs ~60 ms per call at the 96x64x64 training grid; this kernel streams each
// activation once per workgroup tile and keeps ALL 27 taps' accumulators in registers.
//
// Decomposition (persistent): one workgroup per CU, 16 waves; wave (a, b) owns the 16 x 16 block
// (co in [16a,16a+16), ci in [16b,16b+16)) of EVERY tap: 27 x v_mfma_f32_16x16x4_f32 accumulators = 108 VGPRs.
// The workgroup walks its share of 2 x 4 x 16-voxel tiles; per tile the (4 x 6 x 18)-voxel halo of X (all 64
// input channels, 80-float voxel stride: the 2 x 32-lane halves of a ds_read_b32 hit 32 distinct banks) is
// staged in LDS, and each step contracts 4 consecutive voxels: A = dY (global, 64-B rows), B = X at the 27 tap
// offsets (LDS).  Partials [workgroup][27][64][Cin] are reduced by a second small kernel (no atomics:
// bitwise reproducible).
#include "common.hpp"

namespace nrgbd {

typedef float f32x4w __attribute__((ext_vector_type(4)));

constexpr int kWD = 2, kWH = 4, kWW = 16;                       // voxel tile
constexpr int kWHD = kWD + 2, kWHH = kWH + 2, kWHW = kWW + 2;   // halo tile
constexpr int kWHalo = kWHD * kWHH * kWHW;                      // 432
constexpr int kWSV = 80;                                        // LDS voxel stride (floats)

struct WgradArgs {
    const float* x;    // [D][H][W][Cin]  conv input (activated)
    const float* gy;   // [D][H][W][64]   gradient w.r.t. the conv output
    float* partial;    // [gridDim.x][27][64][Cin]
    int D, H, W, Cin;
};

__global__ __launch_bounds__(1024, 4) void conv3d_wgrad_kernel(const WgradArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];  // [kWHalo][kWSV]
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int nci = a.Cin >> 4;                       // ci blocks (1 or 4)
    const int cob = wv & 3, cib = wv >> 2;            // this wave's (co, ci) block
    const bool wave_on = cib < nci;
    const int i16 = lane & 15, k4 = lane >> 4;        // MFMA row/col (0..15) and k (0..3)

    f32x4w acc[27];
#pragma unroll
    for (int t = 0; t < 27; ++t) acc[t] = f32x4w{0.f, 0.f, 0.f, 0.f};

    const int tiles_x = (a.W + kWW - 1) / kWW, tiles_y = (a.H + kWH - 1) / kWH, tiles_z = (a.D + kWD - 1) / kWD;
    const int ntiles = tiles_x * tiles_y * tiles_z;
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        int t = tile;
        const int tx = t % tiles_x; t /= tiles_x;
        const int ty = t % tiles_y; const int tz = t / tiles_y;
        const int x0 = tx * kWW, y0 = ty * kWH, z0 = tz * kWD;
        __syncthreads();  // previous tile's readers are done
        // ---- stage the halo tile of X, all Cin channels, zero outside the volume ----
        const int c4n = a.Cin >> 2;
        for (int idx = tid; idx < kWHalo * c4n; idx += 1024) {
            const int hv = idx / c4n, c4 = idx - hv * c4n;
            const int hz = hv / (kWHH * kWHW), rem = hv - hz * (kWHH * kWHW);
            const int hy = rem / kWHW, hx = rem - hy * kWHW;
            const int gz = z0 + hz - 1, gy_ = y0 + hy - 1, gx = x0 + hx - 1;
            f32x4w v = {0.f, 0.f, 0.f, 0.f};
            if (gz >= 0 && gz < a.D && gy_ >= 0 && gy_ < a.H && gx >= 0 && gx < a.W)
                v = *reinterpret_cast<const f32x4w*>(a.x + (((size_t)gz * a.H + gy_) * a.W + gx) * a.Cin + c4 * 4);
            *reinterpret_cast<f32x4w*>(lds + hv * kWSV + c4 * 4) = v;
        }
        __syncthreads();
        if (!wave_on) continue;
        // ---- 32 steps of 4 consecutive voxels (along x) ----
#pragma unroll 1
        for (int step = 0; step < (kWD * kWH * kWW) / 4; ++step) {
            const int vz = step / (kWH * kWW / 4), r2 = step - vz * (kWH * kWW / 4);
            const int vy = r2 / (kWW / 4), vx = (r2 - vy * (kWW / 4)) * 4 + k4;   // this lane's voxel (k = lane>>4)
            const int gz = z0 + vz, gy_ = y0 + vy, gx = x0 + vx;
            float av = 0.f;   // A[i = co][k = voxel]
            if (gz < a.D && gy_ < a.H && gx < a.W)
                av = a.gy[(((size_t)gz * a.H + gy_) * a.W + gx) * 64 + cob * 16 + i16];
            const float* bbase = lds + ((vz * kWHH + vy) * kWHW + vx) * kWSV + cib * 16 + i16;  // B[k = voxel][j = ci], tap (0,0,0)
#pragma unroll
            for (int tap = 0; tap < 27; ++tap) {
                const int kd = tap / 9, kh = (tap / 3) % 3, kw = tap % 3;
                const float bv = bbase[((kd * kWHH + kh) * kWHW + kw) * kWSV];
                acc[tap] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc[tap], 0, 0, 0);
            }
        }
    }
    if (wave_on) {
        // C/D layout of 16x16x4: col = lane & 15 (j = ci), row = (lane >> 4) * 4 + reg (i = co)
        float* out = a.partial + (size_t)blockIdx.x * 27 * 64 * a.Cin;
#pragma unroll
        for (int tap = 0; tap < 27; ++tap)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int co = cob * 16 + k4 * 4 + r, ci = cib * 16 + i16;
                out[((size_t)tap * 64 + co) * a.Cin + ci] = acc[tap][r];
            }
    }
}

// dW[co][ci][tap] (torch layout [64][Cin][27]) = sum over workgroups of partial[wg][tap][co][ci].  Workgroup = 32 outputs
// (4 consecutive ci each, 16-byte loads) x 8 interleaved slices of the partial list, combined through LDS in index order.
__global__ __launch_bounds__(256) void conv3d_wgrad_reduce_kernel(const float* __restrict__ partial, float* __restrict__ dw,
                                                                  int nwg, int Cin) {
    __shared__ float4 part[8][32];
    const int n = 27 * 64 * Cin, n4 = n >> 2;
    const int lane = threadIdx.x & 31, sl = threadIdx.x >> 5;
    const int idx4 = blockIdx.x * 32 + lane;
    const bool live = idx4 < n4;
    const int idx = live ? idx4 * 4 : 0;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    if (live) {
#pragma unroll 4
        for (int g = sl; g < nwg; g += 8) {
            const float4 q = *reinterpret_cast<const float4*>(partial + (size_t)g * n + idx);
            s.x += q.x; s.y += q.y; s.z += q.z; s.w += q.w;
        }
    }
    part[sl][lane] = s;
    __syncthreads();
    if (sl == 0 && live) {
#pragma unroll
        for (int q = 1; q < 8; ++q) { s.x += part[q][lane].x; s.y += part[q][lane].y; s.z += part[q][lane].z; s.w += part[q][lane].w; }
        const int ci = idx % Cin, co = (idx / Cin) % 64, tap = idx / (Cin * 64);
        float* o = dw + ((size_t)co * Cin + ci) * 27 + tap;
        o[0] = s.x; o[27] = s.y; o[54] = s.z; o[81] = s.w;
    }
}

}  // namespace nrgbd

extern "C" int nrgbd_conv3d_wgrad_workgroups(void) { return 256; }

extern "C" int nrgbd_conv3d_wgrad_f32(const float* x, const float* gy, float* partial, float* dw, int D, int H,
                                      int W, int Cin, void* stream) {
    using namespace nrgbd;
    if (!x || !gy || !partial || !dw) return NRGBD_E_NULL;
    if (D <= 0 || H <= 0 || W <= 0 || (Cin != 16 && Cin != 64)) return NRGBD_E_SHAPE;
    WgradArgs a{x, gy, partial, D, H, W, Cin};
    const int nwg = 256;
    const size_t lds = (size_t)kWHalo * kWSV * sizeof(float);  // 138,240 B
    static_assert((size_t)kWHalo * kWSV * sizeof(float) <= 160 * 1024, "halo tile must fit the 160 KB LDS");
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3d_wgrad_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(conv3d_wgrad_kernel, dim3(nwg), dim3(1024), lds, (hipStream_t)stream, a);
    hipLaunchKernelGGL(conv3d_wgrad_reduce_kernel, dim3(ceil_div(27 * 64 * Cin / 4, 32)), dim3(256), 0, (hipStream_t)stream,
                       partial, dw, nwg, Cin);
    NRGBD_CHECK_LAUNCH();
    return NRGBD_OK;
}
