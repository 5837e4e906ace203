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

// Round 3: 8 waves of 256 registers instead of 16 of 128.  Wave (p, b) owns the ci block b of BOTH co blocks 2p, 2p+1 of every
// tap (216 accumulator registers): a B operand read from LDS feeds two MFMAs, there is room to keep several LDS reads in flight
// (the 128-register version spilled and waited for every single ds_read before its MFMA), and the dY elements of step s + 1 are
// requested — unconditionally, from a clamped address, so that the compiler can count them — before the 54 MFMAs of step s issue.
constexpr int kWThreads = 512;
__device__ float g_wgrad_zeros[64];   // what lanes whose voxel lies outside the volume load instead of dY (never written)

__global__ __launch_bounds__(kWThreads) void conv3d_wgrad_kernel(const WgradArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];  // [kWHalo][kWSV]
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int nci = a.Cin >> 4;                       // ci blocks (1 or 4)
    const int cib = wv & 3, cop = wv >> 2;            // this wave's ci block and pair of co blocks
    const bool wave_on = cib < nci;
    const int i16 = lane & 15, k4 = lane >> 4;        // MFMA row/col (0..15) and k (0..3)

    f32x4w acc[2][27];
#pragma unroll
    for (int g = 0; g < 2; ++g)
#pragma unroll
        for (int t = 0; t < 27; ++t) acc[g][t] = f32x4w{0.f, 0.f, 0.f, 0.f};

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
        for (int idx = tid; idx < kWHalo * c4n; idx += kWThreads) {
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
        // A[i = co][k = voxel] of a step for the wave's two co blocks: an unconditional load (lanes outside the volume read zeros)
        auto load_a = [&](int step, float (&av)[2]) {
            const int vz = step / (kWH * kWW / 4), r2 = step - vz * (kWH * kWW / 4);
            const int vy = r2 / (kWW / 4), vx = (r2 - vy * (kWW / 4)) * 4 + k4;   // this lane's voxel (k = lane>>4)
            const int gz = z0 + vz, gy_ = y0 + vy, gx = x0 + vx;
            const bool ok = gz < a.D && gy_ < a.H && gx < a.W;
            const float* p = a.gy + (((size_t)gz * a.H + gy_) * a.W + gx) * 64 + cop * 32 + i16;
            p = ok ? p : g_wgrad_zeros + i16;          // the select is on the address: nothing depends on the loaded value but the MFMAs
            av[0] = p[0]; av[1] = p[16];
        };
        auto step_mfma = [&](int step, const float (&av)[2]) {
            const int vz = step / (kWH * kWW / 4), r2 = step - vz * (kWH * kWW / 4);
            const int vy = r2 / (kWW / 4), vx = (r2 - vy * (kWW / 4)) * 4 + k4;
            const float* bbase = lds + ((vz * kWHH + vy) * kWHW + vx) * kWSV + cib * 16 + i16;  // B[k = voxel][j = ci], tap (0,0,0)
#pragma unroll
            for (int tap = 0; tap < 27; ++tap) {
                const int kd = tap / 9, kh = (tap / 3) % 3, kw = tap % 3;
                const float bv = bbase[((kd * kWHH + kh) * kWHW + kw) * kWSV];
                acc[0][tap] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[0], bv, acc[0][tap], 0, 0, 0);
                acc[1][tap] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[1], bv, acc[1][tap], 0, 0, 0);
            }
        };
        constexpr int NSTEP = (kWD * kWH * kWW) / 4;
        float a0[2], a1[2];
        load_a(0, a0);
#pragma unroll 1
        for (int step = 0; step < NSTEP; step += 2) {      // two steps per trip: the operand registers alternate, nothing is moved
            load_a(step + 1, a1);
            __builtin_amdgcn_sched_barrier(0);             // the requests leave BEFORE the MFMAs of this step (the scheduler sinks them otherwise)
            step_mfma(step, a0);
            __builtin_amdgcn_sched_barrier(0);
            load_a(min(step + 2, NSTEP - 1), a0);
            __builtin_amdgcn_sched_barrier(0);
            step_mfma(step + 1, a1);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    if (wave_on) {
        // C/D layout of 16x16x4: col = lane & 15 (j = ci), row = (lane >> 4) * 4 + reg (i = co)
        float* out = a.partial + (size_t)blockIdx.x * 27 * 64 * a.Cin;
#pragma unroll
        for (int g = 0; g < 2; ++g)
#pragma unroll
            for (int tap = 0; tap < 27; ++tap)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int co = (2 * cop + g) * 16 + k4 * 4 + r, ci = cib * 16 + i16;
                    out[((size_t)tap * 64 + co) * a.Cin + ci] = acc[g][tap][r];
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
    hipLaunchKernelGGL(conv3d_wgrad_kernel, dim3(nwg), dim3(kWThreads), lds, (hipStream_t)stream, a);
    hipLaunchKernelGGL(conv3d_wgrad_reduce_kernel, dim3(ceil_div(27 * 64 * Cin / 4, 32)), dim3(256), 0, (hipStream_t)stream,
                       partial, dw, nwg, Cin);
    NRGBD_CHECK_LAUNCH();
    return NRGBD_OK;
}
