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
    extern __shared__ __attribute__((aligned(16))) float lds[];  // [HALO][kGSV]
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
        __syncthreads();
        if (!wave_on) continue;
        // ---- 32 steps of 4 consecutive pixels (along x) ----
#pragma unroll 1
        for (int step = 0; step < (kGH * kGW) / 4; ++step) {
            const int vy = step / (kGW / 4), vx = (step - vy * (kGW / 4)) * 4 + k4;   // this lane's pixel (k = lane >> 4)
            const int gy_ = y0 + vy, gx = x0 + vx;
            float av = 0.f;   // A[i = co][k = pixel]
            if (gy_ < a.H && gx < a.W)
                av = a.gy[(((size_t)n * a.H + gy_) * a.W + gx) * a.Cout + cog * 64 + cob * 16 + i16];
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

// dW[co][ci][tap] (torch layout [Cout][Cin][3][3]) = sum over the block's workgroups of partial[block][wg][tap][co % 64][ci % 64]
__global__ __launch_bounds__(256) void conv2d_wgrad_reduce_kernel(const float* __restrict__ partial, float* __restrict__ dw,
                                                                  int nwg, int Cin, int Cout, int ncig) {
    const long n = (long)9 * Cin * Cout;
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= n) return;
    const int tap = (int)(idx % 9);
    const int ci = (int)((idx / 9) % Cin), co = (int)(idx / (9 * (long)Cin));
    const int blk = (co >> 6) * ncig + (ci >> 6);
    const float* p = partial + (size_t)blk * nwg * (9 * 64 * 64) + ((size_t)tap * 64 + (co & 63)) * 64 + (ci & 63);
    float s = 0.f;
    for (int g = 0; g < nwg; ++g) s += p[(size_t)g * (9 * 64 * 64)];
    dw[idx] = s;   // idx = (co * Cin + ci) * 9 + tap
}

}  // namespace nrgbd

// workgroups per 64 x 64 weight block: enough to fill the chip on large problems, but never fewer than 4 pixel tiles per
// workgroup — every workgroup writes a 147 KB partial that the reduction has to read back (at the 64x96 training grid 256
// single-tile workgroups made the reduction as expensive as the gradient itself)
extern "C" int nrgbd_conv2d_wgrad_workgroups(int N, int H, int W, int Cin, int Cout) {
    if (N <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cin % 16 || Cout <= 0 || Cout % 16) return NRGBD_E_SHAPE;
    const int blocks = ((Cout + 63) / 64) * ((Cin + 63) / 64);
    const long ntiles = (long)N * ((H + nrgbd::kGH - 1) / nrgbd::kGH) * ((W + nrgbd::kGW - 1) / nrgbd::kGW);
    long per = 256 / blocks;
    if (per < 1) per = 1;
    if (per > (ntiles + 3) / 4) per = (ntiles + 3) / 4;
    return (int)(per > 0 ? per : 1);
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
    const size_t lds = (size_t)(kGH + 2 * dilation) * (kGW + 2 * dilation) * kGSV * sizeof(float);   // 57.6 / 76.8 KB
    hipError_t e;
    if (dilation == 1) {
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv2d_wgrad_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
        hipLaunchKernelGGL(conv2d_wgrad_kernel<1>, dim3(nwg, blocks), dim3(1024), lds, (hipStream_t)stream, a);
    } else {
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv2d_wgrad_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
        hipLaunchKernelGGL(conv2d_wgrad_kernel<2>, dim3(nwg, blocks), dim3(1024), lds, (hipStream_t)stream, a);
    }
    const long n = (long)9 * Cin * Cout;
    hipLaunchKernelGGL(conv2d_wgrad_reduce_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, partial, dw,
                       nwg, Cin, Cout, ncig);
    NRGBD_CHECK_LAUNCH();
    return NRGBD_OK;
}
