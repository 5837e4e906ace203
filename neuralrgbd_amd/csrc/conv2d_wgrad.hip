// conv2d_wgrad.hip — weight gradient of the 3x3 stride-1 convolutions of the feature CNN / R-Net on the fp32 matrix cores
// (training: train_utils/train_KVNet.py:103-153 back-propagates through models/psm_submodule.py:10-16,31-50 and
// models/Refine.py:51-107).  The 2-D sibling of conv3d_wgrad.hip:
//
//   dW[co][ci][ky][kx] = sum over (n, y, x) of  dY[n][y][x][co] * X[n][y + (ky-1) d][x + (kx-1) d][ci]     (X zero outside the image)
//
// i.e. 9 skinny GEMMs (Cout x Cin, K = N*H*W pixels) that share their operands.  The weight matrix is cut into 64 x 64
// (co, ci) blocks; blockIdx.y = block, blockIdx.x = one of the persistent workgroups of that block, which walks its share of
// 8 x 16-pixel tiles: the (8+2d) x (16+2d)-pixel halo of X (the block's 64 input channels, 80-float pixel stride: the two
// 32-lane halves of a ds_read_b32 hit 32 distinct banks) is staged in LDS and each step contracts 4 consecutive pixels:
// A = dY (global, 64-byte rows), B = X at the 9 tap offsets (LDS).  Wave (a, b) of the 16 owns the 16 x 16 sub-block
// (co 16a.., ci 16b..) of every tap: 9 x v_mfma_f32_16x16x4_f32 accumulators.  Partials [workgroup][9][64][64] are reduced by a
// second small kernel (no atomics: bitwise reproducible).
// Round 3: the tile of dY is staged in LDS as well (all 1,024 threads, 16-byte coalesced loads) — the first version read a
// lane's dY element straight from global memory inside every 4-pixel step, one exposed memory latency per 9 MFMAs — and a
// workgroup owns ONE pixel tile (the 64x96 training grid has 240 of them per image batch: the first version's "at least four
// tiles per workgroup" left 196 of 256 CUs idle to keep the partials small; the reduction now reads them 16 bytes per thread).
#include "common.hpp"

namespace nrgbd {

typedef float f32x4g __attribute__((ext_vector_type(4)));

constexpr int kGH = 8, kGW = 16;   // pixel tile
constexpr int kGSV = 80;           // LDS pixel stride (floats)

struct Wgrad2dArgs {
    const float* x;    // [N][H][W][Cin]   conv input (activated)
    const float* gy;   // [N][H][W][Cout]  gradient w.r.t. the conv output
    float* partial;    // [blocks][gridDim.x][9][64][64]
    int N, H, W, Cin, Cout, ncig;   // ncig = ceil(Cin / 64)
};

template <int DIL>
__global__ __launch_bounds__(1024) void conv2d_wgrad_kernel(const Wgrad2dArgs a) {
    constexpr int HH = kGH + 2 * DIL, HW = kGW + 2 * DIL, HALO = HH * HW;
    extern __shared__ __attribute__((aligned(16))) float lds[];  // X halo [HALO][kGSV], then the dY tile [kGH*kGW][kGSV]
    float* ldg = lds + HALO * kGSV;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int cog = blockIdx.y / a.ncig, cig = blockIdx.y - cog * a.ncig;
    const int cin0 = cig * 64, nci = min(64, a.Cin - cin0) >> 4;   // 16-channel sub-blocks of this block that exist
    const int nco = min(64, a.Cout - cog * 64) >> 4;
    const int cob = wv & 3, cib = wv >> 2;
    const bool wave_on = cib < nci && cob < nco;
    const int i16 = lane & 15, k4 = lane >> 4;

    f32x4g acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) acc[t] = f32x4g{0.f, 0.f, 0.f, 0.f};

    const int tiles_x = (a.W + kGW - 1) / kGW, tiles_y = (a.H + kGH - 1) / kGH;
    const int ntiles = tiles_x * tiles_y * a.N;
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        int t = tile;
        const int tx = t % tiles_x; t /= tiles_x;
        const int ty = t % tiles_y; const int n = t / tiles_y;
        const int x0 = tx * kGW, y0 = ty * kGH;
        __syncthreads();  // previous tile's readers are done
        // ---- stage the halo tile of X (this block's input channels), zero outside the image ----
        const int c4n = nci * 4;
        for (int idx = tid; idx < HALO * c4n; idx += 1024) {
            const int hv = idx / c4n, c4 = idx - hv * c4n;
            const int hy = hv / HW, hx = hv - hy * HW;
            const int gy_ = y0 + hy - DIL, gx = x0 + hx - DIL;
            f32x4g v = {0.f, 0.f, 0.f, 0.f};
            if (gy_ >= 0 && gy_ < a.H && gx >= 0 && gx < a.W)
                v = *reinterpret_cast<const f32x4g*>(a.x + (((size_t)n * a.H + gy_) * a.W + gx) * a.Cin + cin0 + c4 * 4);
            *reinterpret_cast<f32x4g*>(lds + hv * kGSV + c4 * 4) = v;
        }
        // ---- and the tile of dY (this block's output channels), zero outside the image ----
        const int o4n = nco * 4;
        for (int idx = tid; idx < kGH * kGW * o4n; idx += 1024) {
            const int pv = idx / o4n, c4 = idx - pv * o4n;
            const int vy = pv / kGW, vx = pv - vy * kGW;
            const int gy_ = y0 + vy, gx = x0 + vx;
            f32x4g v = {0.f, 0.f, 0.f, 0.f};
            if (gy_ < a.H && gx < a.W)
                v = *reinterpret_cast<const f32x4g*>(a.gy + (((size_t)n * a.H + gy_) * a.W + gx) * a.Cout + cog * 64 + c4 * 4);
            *reinterpret_cast<f32x4g*>(ldg + pv * kGSV + c4 * 4) = v;
        }
        __syncthreads();
        if (!wave_on) continue;
        // ---- 32 steps of 4 consecutive pixels (along x) ----
#pragma unroll 1
        for (int step = 0; step < (kGH * kGW) / 4; ++step) {
            const int vy = step / (kGW / 4), vx = (step - vy * (kGW / 4)) * 4 + k4;   // this lane's pixel (k = lane >> 4)
            const float av = ldg[(vy * kGW + vx) * kGSV + cob * 16 + i16];   // A[i = co][k = pixel]
            const float* bbase = lds + (vy * HW + vx) * kGSV + cib * 16 + i16;  // B[k = pixel][j = ci], tap (0,0)
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const int kh = tap / 3, kw = tap % 3;
                const float bv = bbase[((kh * HW + kw) * DIL) * kGSV];
                acc[tap] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc[tap], 0, 0, 0);
            }
        }
    }
    if (wave_on) {
        // C/D layout of 16x16x4: col = lane & 15 (j = ci), row = (lane >> 4) * 4 + reg (i = co)
        float* out = a.partial + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 9 * 64 * 64;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int co = cob * 16 + k4 * 4 + r, ci = cib * 16 + i16;
                out[((size_t)tap * 64 + co) * 64 + ci] = acc[tap][r];
            }
    }
}

// dW[co][ci][tap] (torch layout [Cout][Cin][3][3]) = sum over the block's workgroups of partial[block][wg][tap][co % 64][ci % 64].
// Workgroup = 32 outputs (tap, co, 4 consecutive ci: 16-byte loads, coalesced along ci) x 8 interleaved slices of the
// workgroup list; the slices are added through LDS in index order, so the result does not depend on scheduling.
__global__ __launch_bounds__(256) void conv2d_wgrad_reduce_kernel(const float* __restrict__ partial, float* __restrict__ dw,
                                                                  int nwg, int Cin, int Cout, int ncig) {
    __shared__ f32x4g part[8][32];
    const int cin4 = Cin >> 2;
    const long n = (long)9 * cin4 * Cout;
    const int lane = threadIdx.x & 31, sl = threadIdx.x >> 5;
    const long idx = (long)blockIdx.x * 32 + lane;
    const bool live = idx < n;
    const long id = live ? idx : 0;
    const int c4 = (int)(id % cin4);
    const int co = (int)((id / cin4) % Cout), tap = (int)(id / ((long)cin4 * Cout));
    const int ci = c4 * 4;
    const int blk = (co >> 6) * ncig + (ci >> 6);
    const float* p = partial + (size_t)blk * nwg * (9 * 64 * 64) + ((size_t)tap * 64 + (co & 63)) * 64 + (ci & 63);
    f32x4g s4 = {0.f, 0.f, 0.f, 0.f};
    if (live) {
#pragma unroll 4
        for (int g = sl; g < nwg; g += 8) s4 = s4 + *reinterpret_cast<const f32x4g*>(p + (size_t)g * (9 * 64 * 64));
    }
    part[sl][lane] = s4;
    __syncthreads();
    if (sl == 0 && live) {
#pragma unroll
        for (int q = 1; q < 8; ++q) s4 = s4 + part[q][lane];
        float* o = dw + ((size_t)co * Cin + ci) * 9 + tap;
        o[0] = s4.x; o[9] = s4.y; o[18] = s4.z; o[27] = s4.w;
    }
}

}  // namespace nrgbd

// workgroups per 64 x 64 weight block; each writes a 147,456-byte partial (9 taps x 64 x 64 floats, also for narrower layers)
// that the reduction reads back, so the scratch is blocks x workgroups x 147 KB: bounded to ~256 MB here (it used to
// reach 1.5 GB for the 320 -> 128 layer on large grids)
extern "C" int nrgbd_conv2d_wgrad_workgroups(int N, int H, int W, int Cin, int Cout) {
    if (N <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cin % 16 || Cout <= 0 || Cout % 16) return NRGBD_E_SHAPE;
    const int blocks = ((Cout + 63) / 64) * ((Cin + 63) / 64);
    const long ntiles = (long)N * ((H + nrgbd::kGH - 1) / nrgbd::kGH) * ((W + nrgbd::kGW - 1) / nrgbd::kGW);
    // one pixel tile per workgroup while the partials of all blocks stay below ~256 MB (1,820 partials) and 1,024 per block;
    // beyond that the workgroups walk the tile list.  >= 64 per block keeps every CU busy for single-block layers.
    long cap = 1820 / blocks;
    if (cap < 64) cap = 64;
    if (cap > 1024) cap = 1024;
    return (int)(ntiles < cap ? ntiles : cap);
}

extern "C" int nrgbd_conv2d_wgrad_f32(const float* x, const float* gy, float* partial, float* dw, int N, int H, int W, int Cin,
                                      int Cout, int dilation, void* stream) {
    using namespace nrgbd;
    if (!x || !gy || !partial || !dw) return NRGBD_E_NULL;
    if (N <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cin % 16 || Cout <= 0 || Cout % 16) return NRGBD_E_SHAPE;
    if (dilation != 1 && dilation != 2) return NRGBD_E_ARG;
    const int ncig = (Cin + 63) / 64, blocks = ((Cout + 63) / 64) * ncig;
    const int nwg = nrgbd_conv2d_wgrad_workgroups(N, H, W, Cin, Cout);
    Wgrad2dArgs a{x, gy, partial, N, H, W, Cin, Cout, ncig};
    const size_t lds = (size_t)((kGH + 2 * dilation) * (kGW + 2 * dilation) + kGH * kGW) * kGSV * sizeof(float);   // 98.6 / 117.8 KB
    hipError_t e;
    if (dilation == 1) {
        e = set_max_dynamic_lds(reinterpret_cast<const void*>(&conv2d_wgrad_kernel<1>), (int)lds);
        if (e != hipSuccess) return (int)e;
        hipLaunchKernelGGL(conv2d_wgrad_kernel<1>, dim3(nwg, blocks), dim3(1024), lds, (hipStream_t)stream, a);
    } else {
        e = set_max_dynamic_lds(reinterpret_cast<const void*>(&conv2d_wgrad_kernel<2>), (int)lds);
        if (e != hipSuccess) return (int)e;
        hipLaunchKernelGGL(conv2d_wgrad_kernel<2>, dim3(nwg, blocks), dim3(1024), lds, (hipStream_t)stream, a);
    }
    const long n = (long)9 * (Cin / 4) * Cout;
    hipLaunchKernelGGL(conv2d_wgrad_reduce_kernel, dim3((unsigned)((n + 31) / 32)), dim3(256), 0, (hipStream_t)stream, partial, dw,
                       nwg, Cin, Cout, ncig);
    NRGBD_CHECK_LAUNCH();
    return NRGBD_OK;
}
