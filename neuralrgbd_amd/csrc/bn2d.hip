// bn2d.hip — train-mode BatchNorm2d fused with its activation and residual add, and the SPP pooling base,
// for the 2-D feature CNN (psm_submodule.py:10-16 convbn, :31-50 BasicBlock, :103-117 SPP branches).
//
// The reference never leaves train() mode, so every BatchNorm2d normalises with the statistics of the
// current 5-image batch.  MIOpen's BN + separate ReLU + separate residual-add kernels move
// each activation 5-6 times; here it is two HBM-bound passes over NCHW:
//   bn2d_stats : per-channel partial (sum, sum of squares) over slices of N*H*W      (read x once)
//   bn2d_apply : y = act(x*s + t) (+ residual), s/t derived in the prologue (fp64) from the partials
//                                                                                  (read x [+res], write y)
// Both kernels vectorise 16 B per lane along the contiguous H*W axis.
#include "common.hpp"

namespace nrgbd {

constexpr int kBnSplit = 32;  // slices of the (n, hw) range per channel in the statistics pass

// grid (C, kBnSplit), block 256: partial[c][s] = (sum, sumsq) of channel c over slice s
__global__ __launch_bounds__(256) void bn2d_stats_kernel(const float* __restrict__ x, float* __restrict__ partial,
                                                         int N, int C, long HW) {
    const int c = blockIdx.x, s = blockIdx.y, tid = threadIdx.x;
    // slice the per-image HW range in float4 units (HW % 4 == 0 is checked by the host; else scalar path)
    const long hw4 = HW >> 2;
    const long per = (hw4 + kBnSplit - 1) / kBnSplit;
    const long b = (long)s * per, e = min(hw4, b + per);
    float s1 = 0.f, s2 = 0.f;
    for (int n = 0; n < N; ++n) {
        const float4* p = reinterpret_cast<const float4*>(x + ((size_t)n * C + c) * HW);
        for (long i = b + tid; i < e; i += 256) {
            const float4 v = p[i];
            s1 += (v.x + v.y) + (v.z + v.w);
            s2 = __builtin_fmaf(v.x, v.x, s2); s2 = __builtin_fmaf(v.y, v.y, s2);
            s2 = __builtin_fmaf(v.z, v.z, s2); s2 = __builtin_fmaf(v.w, v.w, s2);
        }
        if (s == kBnSplit - 1)  // tail elements when HW % 4 != 0
            for (long i = (hw4 << 2) + tid; i < HW; i += 256) {
                const float v = x[((size_t)n * C + c) * HW + i];
                s1 += v; s2 = __builtin_fmaf(v, v, s2);
            }
    }
    __shared__ float r1[4], r2[4];
    s1 = wave_sum(s1); s2 = wave_sum(s2);
    if ((tid & 63) == 0) { r1[tid >> 6] = s1; r2[tid >> 6] = s2; }
    __syncthreads();
    if (tid == 0) {
        partial[((size_t)c * kBnSplit + s) * 2 + 0] = (r1[0] + r1[1]) + (r1[2] + r1[3]);
        partial[((size_t)c * kBnSplit + s) * 2 + 1] = (r2[0] + r2[1]) + (r2[2] + r2[3]);
    }
}

// grid (chunks, N*C), block 256.  act: 0 none, 1 ReLU.  mean_var [C][2] (optional) receives the batch
// mean and biased variance (for the running-statistics update of the two shortcut norms).
__global__ __launch_bounds__(256) void bn2d_apply_kernel(const float* __restrict__ x, const float* __restrict__ partial,
                                                         const float* __restrict__ gamma, const float* __restrict__ beta,
                                                         float eps, int act, const float* __restrict__ res,
                                                         float* __restrict__ y, float* __restrict__ mean_var,
                                                         int N, int C, long HW) {
    const int nc = blockIdx.y, c = nc % C;
    double d1 = 0.0, d2 = 0.0;
#pragma unroll 8
    for (int s = 0; s < kBnSplit; ++s) {
        d1 += (double)partial[((size_t)c * kBnSplit + s) * 2 + 0];
        d2 += (double)partial[((size_t)c * kBnSplit + s) * 2 + 1];
    }
    const double cnt = (double)N * (double)HW;
    const double mean = d1 / cnt;
    double var = d2 / cnt - mean * mean;
    var = var > 0.0 ? var : 0.0;
    const float sc = gamma[c] * (float)(1.0 / sqrt(var + (double)eps));
    const float sh = beta[c] - (float)mean * sc;
    if (mean_var && blockIdx.x == 0 && nc < C && threadIdx.x == 0) { mean_var[2 * c] = (float)mean; mean_var[2 * c + 1] = (float)var; }

    const size_t base = (size_t)nc * HW;
    const long hw4 = HW >> 2;
    const long stride = (long)gridDim.x * 256;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < hw4; i += stride) {
        float4 v = reinterpret_cast<const float4*>(x + base)[i];
        v.x = __builtin_fmaf(v.x, sc, sh); v.y = __builtin_fmaf(v.y, sc, sh);
        v.z = __builtin_fmaf(v.z, sc, sh); v.w = __builtin_fmaf(v.w, sc, sh);
        if (act == 1) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
        if (res) {
            const float4 r = reinterpret_cast<const float4*>(res + base)[i];
            v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w;
        }
        reinterpret_cast<float4*>(y + base)[i] = v;
    }
    if (blockIdx.x == 0)
        for (long i = (hw4 << 2) + threadIdx.x; i < HW; i += 256) {
            float v = __builtin_fmaf(x[base + i], sc, sh);
            if (act == 1) v = fmaxf(v, 0.f);
            if (res) v += res[base + i];
            y[base + i] = v;
        }
}

// K x K average pooling with stride K (K = 8: the finest SPP window; the coarser ones are pooled from its
// output).  grid (W/K groups, H/K, N*C): one wave-row of lanes per output row segment; each lane sums a
// K-wide, K-tall window with 16-B loads (K % 4 == 0).
template <int K>
__global__ __launch_bounds__(256) void avgpool_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                      int H, int W) {
    const int ow = W / K, oh = H / K;
    const int ox = blockIdx.x * 256 + threadIdx.x, oy = blockIdx.y, nc = blockIdx.z;
    if (ox >= ow) return;
    const float* p = x + ((size_t)nc * H + (size_t)oy * K) * W + (size_t)ox * K;
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < K; ++j) {
#pragma unroll
        for (int i = 0; i < K; i += 4) {
            const float4 v = *reinterpret_cast<const float4*>(p + (size_t)j * W + i);
            s += v.x; s += v.y; s += v.z; s += v.w;  // row-major order, like at::avg_pool2d
        }
    }
    y[((size_t)nc * oh + oy) * ow + ox] = s / (float)(K * K);
}


// y = leaky(x + bias[c], slope) in place on [N][C][HW] — the R-Net's conv2d_leakyRelu / ConvTranspose2d + LeakyReLU
// tail (models/m_submodule.py:18-27,36-45) as one pass instead of the vendor conv's separate bias-add kernel plus a
// LeakyReLU kernel.  slope = 1 gives the plain bias add of the last layer (Refine.py:71).  grid (chunks, N*C).
__global__ __launch_bounds__(256) void bias_act_nchw_kernel(float* __restrict__ x, const float* __restrict__ bias, float slope,
                                                            int C, long HW) {
    const int nc = blockIdx.y;
    const float b = bias[nc % C];
    const size_t base = (size_t)nc * HW;
    const long hw4 = HW >> 2;
    const long stride = (long)gridDim.x * 256;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < hw4; i += stride) {
        float4 v = reinterpret_cast<float4*>(x + base)[i];
        v.x += b; v.y += b; v.z += b; v.w += b;
        v.x = v.x > 0.f ? v.x : slope * v.x; v.y = v.y > 0.f ? v.y : slope * v.y;
        v.z = v.z > 0.f ? v.z : slope * v.z; v.w = v.w > 0.f ? v.w : slope * v.w;
        reinterpret_cast<float4*>(x + base)[i] = v;
    }
}

}  // namespace nrgbd

extern "C" int nrgbd_bn2d_partial_floats(int C) { return C > 0 ? C * nrgbd::kBnSplit * 2 : NRGBD_E_SHAPE; }

extern "C" int nrgbd_bn2d_train_act(const float* x, const float* gamma, const float* beta, float eps, int act,
                                    const float* residual, float* y, float* partial, float* mean_var,
                                    int N, int C, long HW, void* stream) {
    using namespace nrgbd;
    if (!x || !gamma || !beta || !y || !partial) return NRGBD_E_NULL;
    if (N <= 0 || C <= 0 || HW <= 0 || (long)N * C > 65535) return NRGBD_E_SHAPE;
    if (act != 0 && act != 1) return NRGBD_E_ARG;
    if ((HW & 3) && (reinterpret_cast<uintptr_t>(x) & 15)) return NRGBD_E_ALIGN;
    if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(residual)) & 15) return NRGBD_E_ALIGN;
    if (HW & 3) return NRGBD_E_ALIGN;  // planes must keep 16-B alignment (all maps of the feature CNN do)
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(bn2d_stats_kernel, dim3(C, kBnSplit), dim3(256), 0, s, x, partial, N, C, HW);
    long want = (HW / 4 + 1023) / 1024;
    const int chunks = (int)(want < 1 ? 1 : (want > 64 ? 64 : want));
    hipLaunchKernelGGL(bn2d_apply_kernel, dim3(chunks, N * C), dim3(256), 0, s, x, partial, gamma, beta, eps, act,
                       residual, y, mean_var, N, C, HW);
    NRGBD_CHECK_LAUNCH();
    return NRGBD_OK;
}

extern "C" int nrgbd_avgpool8(const float* x, float* y, int NC, int H, int W, void* stream) {
    using namespace nrgbd;
    if (!x || !y) return NRGBD_E_NULL;
    if (NC <= 0 || NC > 65535 || H < 8 || W < 8 || (W & 3) || H / 8 > 65535) return NRGBD_E_SHAPE;
    if (reinterpret_cast<uintptr_t>(x) & 15) return NRGBD_E_ALIGN;
    dim3 grid(ceil_div(W / 8, 256), H / 8, NC);
    hipLaunchKernelGGL(avgpool_kernel<8>, grid, dim3(256), 0, (hipStream_t)stream, x, y, H, W);
    NRGBD_CHECK_LAUNCH();
    return NRGBD_OK;
}

extern "C" int nrgbd_bias_act_nchw(float* x, const float* bias, float slope, int N, int C, long HW, void* stream) {
    using namespace nrgbd;
    if (!x || !bias) return NRGBD_E_NULL;
    if (N <= 0 || C <= 0 || HW <= 0 || (HW & 3) || (long)N * C > 65535) return NRGBD_E_SHAPE;
    if (reinterpret_cast<uintptr_t>(x) & 15) return NRGBD_E_ALIGN;
    long want = (HW / 4 + 1023) / 1024;
    const int chunks = (int)(want < 1 ? 1 : (want > 256 ? 256 : want));
    hipLaunchKernelGGL(bias_act_nchw_kernel, dim3(chunks, N * C), dim3(256), 0, (hipStream_t)stream, x, bias, slope, C, HW);
    NRGBD_CHECK_LAUNCH();
    return NRGBD_OK;
}
