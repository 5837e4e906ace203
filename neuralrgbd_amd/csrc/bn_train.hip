// bn_train.hip — train-mode BatchNorm (batch statistics) with its activation and residual add, forward AND backward, on
// channels-last activations viewed as [rows][C] (rows = N*H*W for the feature CNN's BatchNorm2d, D*H*W for the K-Net's
// BatchNorm3d).  Training path only (BASELINE config 4): what autograd records in the reference for
//   models/psm_submodule.py:10-16 (convbn), :31-50 (BasicBlock: relu after the first norm, `out += x` after the second),
//   models/basic.py:53-68 (convbn_3d), :71-94 (KV_NET_BASIC: relu, residual adds),  train_utils/train_KVNet.py:152 (backward).
// All six kernels are HBM-bound streaming passes, 16 bytes per lane along C:
//   forward   stats  (read x)            partial (sum, sum of squares) per workgroup
//             finalize                   fp64 over the partials -> coef [4][C] = scale, shift, mean, invstd (+ running statistics)
//             apply  (read x [,res], write y)     y = act(x*scale + shift) + res
//   backward  reduce (read x, gy)        partial (sum dz, sum dz*(x - mean)),  dz = gy masked by the ReLU (recomputed from x)
//             finalize                   g_gamma, g_beta, coef2 [2][C] = B, K
//             apply  (read x, gy, write gx)       gx = scale*dz + B*(x - mean) + K
// with B = -scale*invstd^2 * sum(dz*(x-mean)) / rows and K = -scale * sum(dz) / rows — the closed form of
// d/dx [gamma * (x - mean(x)) * invstd(x) + beta].  The partial sums are added in index order (no atomics).
#include "common.hpp"

namespace nrgbd {

constexpr int kBnclThreads = 256;

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }

// grid G, block 256: lane = (row slot, channel quad); rows walk with stride G * slots.  BWD: the two backward sums.
template <bool BWD>
__global__ __launch_bounds__(kBnclThreads) void bn_cl_stats_kernel(const float* __restrict__ x, const float* __restrict__ gy,
                                                                   const float* __restrict__ coef, int relu,
                                                                   float* __restrict__ partial, long rows, int C) {
    __shared__ float4 sh1[kBnclThreads], sh2[kBnclThreads];
    const int Q = C >> 2, S = kBnclThreads / Q, tid = threadIdx.x;
    const int q = tid % Q, slot = tid / Q;
    float4 sc = make_float4(0.f, 0.f, 0.f, 0.f), sf = sc, mu = sc;
    if (BWD) { sc = ld4(coef + 4 * q); sf = ld4(coef + C + 4 * q); mu = ld4(coef + 2 * C + 4 * q); }
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = a;
    // forward: sums of (x - k) and (x - k)^2 with k = the channel's value in row 0 (the same for every workgroup): the variance
    // E[(x-k)^2] - E[x-k]^2 then has no cancellation when |mean| >> std (a sample of the channel is within a few std of its
    // mean), which E[x^2] - mean^2 on fp32 sums has (torch's kernel is Welford)
    const float4 k = BWD ? a : ld4(x + 4 * q);
    const long step = (long)gridDim.x * S;
#pragma unroll 4
    for (long r = (long)blockIdx.x * S + slot; r < rows; r += step) {
        const float4 v = ld4(x + r * C + 4 * q);
        if (!BWD) {
            const float4 d = make_float4(v.x - k.x, v.y - k.y, v.z - k.z, v.w - k.w);
            a.x += d.x; a.y += d.y; a.z += d.z; a.w += d.w;
            b.x = __builtin_fmaf(d.x, d.x, b.x); b.y = __builtin_fmaf(d.y, d.y, b.y);
            b.z = __builtin_fmaf(d.z, d.z, b.z); b.w = __builtin_fmaf(d.w, d.w, b.w);
        } else {
            float4 g = ld4(gy + r * C + 4 * q);
            if (relu) {
                g.x = __builtin_fmaf(v.x, sc.x, sf.x) > 0.f ? g.x : 0.f; g.y = __builtin_fmaf(v.y, sc.y, sf.y) > 0.f ? g.y : 0.f;
                g.z = __builtin_fmaf(v.z, sc.z, sf.z) > 0.f ? g.z : 0.f; g.w = __builtin_fmaf(v.w, sc.w, sf.w) > 0.f ? g.w : 0.f;
            }
            a.x += g.x; a.y += g.y; a.z += g.z; a.w += g.w;
            b.x = __builtin_fmaf(g.x, v.x - mu.x, b.x); b.y = __builtin_fmaf(g.y, v.y - mu.y, b.y);
            b.z = __builtin_fmaf(g.z, v.z - mu.z, b.z); b.w = __builtin_fmaf(g.w, v.w - mu.w, b.w);
        }
    }
    sh1[tid] = a; sh2[tid] = b;
    __syncthreads();
    if (tid < Q) {
        for (int s = 1; s < S; ++s) {
            const float4 u = sh1[s * Q + tid], w = sh2[s * Q + tid];
            a.x += u.x; a.y += u.y; a.z += u.z; a.w += u.w;
            b.x += w.x; b.y += w.y; b.z += w.z; b.w += w.w;
        }
        float* o = partial + (size_t)blockIdx.x * 2 * C + 4 * tid;
        *reinterpret_cast<float4*>(o) = a;
        *reinterpret_cast<float4*>(o + C) = b;
    }
}

// grid ceil(C/16), block 256 = 16 channels x 16 interleaved slices of the partial list, added in double in a fixed order.
// BWD == false: coef [4][C] = scale, shift, mean, invstd and the running-statistics update (momentum, unbiased variance).
// BWD == true : g_gamma, g_beta, coef2 [2][C] = B, K.
template <bool BWD>
__global__ __launch_bounds__(kBnclThreads) void bn_cl_finalize_kernel(const float* __restrict__ partial, int G, int C, double count,
                                                                      const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                      float eps, float momentum, float* __restrict__ running_mean,
                                                                      float* __restrict__ running_var, float* __restrict__ coef,
                                                                      float* __restrict__ g_gamma, float* __restrict__ g_beta,
                                                                      float* __restrict__ coef2, const float* __restrict__ shift_row) {
    __shared__ double sh[2][16][16];
    const int lane = threadIdx.x & 15, sl = threadIdx.x >> 4;
    const int c = blockIdx.x * 16 + lane;
    double s1 = 0.0, s2 = 0.0;
    if (c < C) {
#pragma unroll 4
        for (int g = sl; g < G; g += 16) {
            s1 += (double)partial[(size_t)g * 2 * C + c];
            s2 += (double)partial[(size_t)g * 2 * C + C + c];
        }
    }
    sh[0][sl][lane] = s1; sh[1][sl][lane] = s2;
    __syncthreads();
    if (sl != 0 || c >= C) return;
#pragma unroll
    for (int q = 1; q < 16; ++q) { s1 += sh[0][q][lane]; s2 += sh[1][q][lane]; }
    if (!BWD) {
        const double dm = s1 / count;                       // mean of (x - k), k = shift_row[c] (row 0 of x)
        const double mean = (double)shift_row[c] + dm;
        double var = s2 / count - dm * dm;
        var = var > 0.0 ? var : 0.0;
        const float invstd = (float)(1.0 / sqrt(var + (double)eps));
        const float scale = gamma[c] * invstd;
        coef[c] = scale;
        coef[C + c] = beta[c] - (float)mean * scale;
        coef[2 * C + c] = (float)mean;
        coef[3 * C + c] = invstd;
        if (running_mean) {
            running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)mean;
            const double unbiased = count > 1.0 ? var * count / (count - 1.0) : var;
            running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unbiased;
        }
    } else {
        const double scale = (double)coef[c], invstd = (double)coef[3 * C + c];
        g_beta[c] = (float)s1;
        g_gamma[c] = (float)(s2 * invstd);
        coef2[c] = (float)(-scale * invstd * invstd * s2 / count);
        coef2[C + c] = (float)(-scale * s1 / count);
    }
}

// one lane = 4 channels of one row.  Forward: y = act(x*scale + shift) + res.
__global__ __launch_bounds__(kBnclThreads) void bn_cl_apply_kernel(const float* __restrict__ x, const float* __restrict__ res,
                                                                   const float* __restrict__ coef, int relu, float* __restrict__ y,
                                                                   long n4, int C) {
    const long i = (long)blockIdx.x * kBnclThreads + threadIdx.x;
    if (i >= n4) return;
    const int q = (int)(i % (C >> 2));
    const float4 sc = ld4(coef + 4 * q), sf = ld4(coef + C + 4 * q);
    const float4 v = ld4(x + 4 * i);
    float4 o = make_float4(__builtin_fmaf(v.x, sc.x, sf.x), __builtin_fmaf(v.y, sc.y, sf.y), __builtin_fmaf(v.z, sc.z, sf.z),
                           __builtin_fmaf(v.w, sc.w, sf.w));
    if (relu) { o.x = o.x > 0.f ? o.x : 0.f; o.y = o.y > 0.f ? o.y : 0.f; o.z = o.z > 0.f ? o.z : 0.f; o.w = o.w > 0.f ? o.w : 0.f; }
    if (res) { const float4 r = ld4(res + 4 * i); o.x += r.x; o.y += r.y; o.z += r.z; o.w += r.w; }
    *reinterpret_cast<float4*>(y + 4 * i) = o;
}

// gx = scale*dz + B*(x - mean) + K
__global__ __launch_bounds__(kBnclThreads) void bn_cl_bwd_apply_kernel(const float* __restrict__ x, const float* __restrict__ gy,
                                                                       const float* __restrict__ coef, const float* __restrict__ coef2,
                                                                       int relu, float* __restrict__ gx, long n4, int C) {
    const long i = (long)blockIdx.x * kBnclThreads + threadIdx.x;
    if (i >= n4) return;
    const int q = (int)(i % (C >> 2));
    const float4 sc = ld4(coef + 4 * q), sf = ld4(coef + C + 4 * q), mu = ld4(coef + 2 * C + 4 * q);
    const float4 B = ld4(coef2 + 4 * q), K = ld4(coef2 + C + 4 * q);
    const float4 v = ld4(x + 4 * i);
    float4 g = ld4(gy + 4 * i);
    if (relu) {
        g.x = __builtin_fmaf(v.x, sc.x, sf.x) > 0.f ? g.x : 0.f; g.y = __builtin_fmaf(v.y, sc.y, sf.y) > 0.f ? g.y : 0.f;
        g.z = __builtin_fmaf(v.z, sc.z, sf.z) > 0.f ? g.z : 0.f; g.w = __builtin_fmaf(v.w, sc.w, sf.w) > 0.f ? g.w : 0.f;
    }
    float4 o;
    o.x = __builtin_fmaf(B.x, v.x - mu.x, __builtin_fmaf(sc.x, g.x, K.x));
    o.y = __builtin_fmaf(B.y, v.y - mu.y, __builtin_fmaf(sc.y, g.y, K.y));
    o.z = __builtin_fmaf(B.z, v.z - mu.z, __builtin_fmaf(sc.z, g.z, K.z));
    o.w = __builtin_fmaf(B.w, v.w - mu.w, __builtin_fmaf(sc.w, g.w, K.w));
    *reinterpret_cast<float4*>(gx + 4 * i) = o;
}

// ---- bias + LeakyReLU of the R-Net's conv2d_leakyRelu / conv2dTranspose_leakyRelu blocks (models/m_submodule.py:18-27,36-45)
// under autograd, channels-last [rows][C]: ONE pass forward (y = lrelu(x + b)), ONE pass backward (gx = gy * lrelu'(y), the bias
// gradient's per-workgroup partial sums in the same pass) instead of torch's add / leaky_relu / leaky_relu_backward / sum.
// slope = 1: the plain bias add of Refine.py:71.  Lane = (row slot, channel quad): Q = C / 4 quads, S = 256 / Q slots (floor:
// widths like 96 leave 256 - S Q lanes idle).
__global__ __launch_bounds__(kBnclThreads) void bias_lrelu_fwd_kernel(const float* __restrict__ x, const float* __restrict__ bias,
                                                                      float slope, float* __restrict__ y, long n4, int C) {
    const long i = (long)blockIdx.x * kBnclThreads + threadIdx.x;
    if (i >= n4) return;
    const int q = (int)(i % (C >> 2));
    const float4 b = ld4(bias + 4 * q), v = ld4(x + 4 * i);
    float4 o = make_float4(v.x + b.x, v.y + b.y, v.z + b.z, v.w + b.w);
    o.x = o.x > 0.f ? o.x : o.x * slope; o.y = o.y > 0.f ? o.y : o.y * slope;
    o.z = o.z > 0.f ? o.z : o.z * slope; o.w = o.w > 0.f ? o.w : o.w * slope;
    *reinterpret_cast<float4*>(y + 4 * i) = o;
}

__global__ __launch_bounds__(kBnclThreads) void bias_lrelu_bwd_kernel(const float* __restrict__ y, const float* __restrict__ gy,
                                                                      float slope, float* __restrict__ gx, float* __restrict__ partial,
                                                                      long rows, int C) {
    __shared__ float4 sh1[kBnclThreads];
    const int Q = C >> 2, S = kBnclThreads / Q, tid = threadIdx.x;
    const int q = tid % Q, slot = tid / Q;
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
    if (slot < S) {
        const long step = (long)gridDim.x * S;
#pragma unroll 4
        for (long r = (long)blockIdx.x * S + slot; r < rows; r += step) {
            const float4 v = ld4(y + r * C + 4 * q);
            float4 g = ld4(gy + r * C + 4 * q);
            g.x = v.x > 0.f ? g.x : g.x * slope; g.y = v.y > 0.f ? g.y : g.y * slope;
            g.z = v.z > 0.f ? g.z : g.z * slope; g.w = v.w > 0.f ? g.w : g.w * slope;
            *reinterpret_cast<float4*>(gx + r * C + 4 * q) = g;
            a.x += g.x; a.y += g.y; a.z += g.z; a.w += g.w;
        }
    }
    sh1[tid] = a;
    __syncthreads();
    if (tid < Q) {
        for (int s = 1; s < S; ++s) {
            const float4 u = sh1[s * Q + tid];
            a.x += u.x; a.y += u.y; a.z += u.z; a.w += u.w;
        }
        *reinterpret_cast<float4*>(partial + (size_t)blockIdx.x * C + 4 * tid) = a;
    }
}

// g_bias[c] = sum over the workgroups' partials, in double, fixed order: a workgroup owns 16 channels, its 16 row slots walk
// the partial list 16 apart and a fixed-order LDS pass adds the slots (one thread per channel walking all G <= 512 partials
// was a 512-deep dependent chain of loads: 40 us per layer)
__global__ __launch_bounds__(kBnclThreads) void bias_lrelu_finalize_kernel(const float* __restrict__ partial, int G, int C,
                                                                           float* __restrict__ g_bias) {
    __shared__ double sh[kBnclThreads];
    const int tid = threadIdx.x, cc = tid & 15, slot = tid >> 4;
    const int c = blockIdx.x * 16 + cc;
    double s = 0.0;
    if (c < C)
        for (int g = slot; g < G; g += kBnclThreads / 16) s += (double)partial[(size_t)g * C + c];
    sh[tid] = s;
    __syncthreads();
    if (tid < 16 && c < C) {
        double t = sh[tid];
        for (int k = 1; k < kBnclThreads / 16; ++k) t += sh[k * 16 + tid];
        g_bias[c] = (float)t;
    }
}

static int bias_lrelu_groups(long rows, int C) {
    const int S = kBnclThreads / (C >> 2);
    const long g = (rows + (long)S * 24 - 1) / ((long)S * 24);
    return (int)(g < 1 ? 1 : (g > 512 ? 512 : g));
}

static bool bn_cl_shape_ok(long rows, int C) {
    return rows > 0 && C >= 4 && C <= 1024 && (C & 3) == 0 && kBnclThreads % (C >> 2) == 0;
}

}  // namespace nrgbd

// Workgroups of the statistics passes = rows of the `partial` scratch ([workgroups][2*C] floats): about 24 rows per lane,
// at most 256 workgroups (one per CU: the finaliser walks the list).
extern "C" int nrgbd_bn_cl_workgroups(long rows, int C) {
    using namespace nrgbd;
    if (!bn_cl_shape_ok(rows, C)) return NRGBD_E_SHAPE;
    const int S = kBnclThreads / (C >> 2);
    const long g = (rows + (long)S * 24 - 1) / ((long)S * 24);
    return (int)(g < 1 ? 1 : (g > 256 ? 256 : g));
}

extern "C" int nrgbd_bn_cl_fwd(const float* x, const float* res, const float* gamma, const float* beta, float eps, float momentum,
                               float* running_mean, float* running_var, int relu, float* y, float* coef, float* partial,
                               long rows, int C, void* stream) {
    using namespace nrgbd;
    if (!x || !gamma || !beta || !y || !coef || !partial) return NRGBD_E_NULL;
    if ((running_mean == nullptr) != (running_var == nullptr)) return NRGBD_E_NULL;
    if (!bn_cl_shape_ok(rows, C)) return NRGBD_E_SHAPE;
    hipStream_t s = (hipStream_t)stream;
    const int G = nrgbd_bn_cl_workgroups(rows, C);
    hipLaunchKernelGGL(bn_cl_stats_kernel<false>, dim3(G), dim3(kBnclThreads), 0, s, x, nullptr, nullptr, 0, partial, rows, C);
    hipLaunchKernelGGL(bn_cl_finalize_kernel<false>, dim3((C + 15) / 16), dim3(kBnclThreads), 0, s, partial, G, C, (double)rows, gamma,
                       beta, eps, momentum, running_mean, running_var, coef, nullptr, nullptr, nullptr, x);
    const long n4 = rows * (C >> 2);
    hipLaunchKernelGGL(bn_cl_apply_kernel, dim3((unsigned)ceil_div(n4, (long)kBnclThreads)), dim3(kBnclThreads), 0, s, x, res, coef, relu,
                       y, n4, C);
    NRGBD_CHECK_LAUNCH();
    return NRGBD_OK;
}

extern "C" int nrgbd_bn_cl_bwd(const float* x, const float* gy, const float* coef, int relu, float* gx, float* g_gamma,
                               float* g_beta, float* coef2, float* partial, long rows, int C, void* stream) {
    using namespace nrgbd;
    if (!x || !gy || !coef || !gx || !g_gamma || !g_beta || !coef2 || !partial) return NRGBD_E_NULL;
    if (!bn_cl_shape_ok(rows, C)) return NRGBD_E_SHAPE;
    hipStream_t s = (hipStream_t)stream;
    const int G = nrgbd_bn_cl_workgroups(rows, C);
    hipLaunchKernelGGL(bn_cl_stats_kernel<true>, dim3(G), dim3(kBnclThreads), 0, s, x, gy, coef, relu, partial, rows, C);
    hipLaunchKernelGGL(bn_cl_finalize_kernel<true>, dim3((C + 15) / 16), dim3(kBnclThreads), 0, s, partial, G, C, (double)rows, nullptr,
                       nullptr, 0.f, 0.f, nullptr, nullptr, const_cast<float*>(coef), g_gamma, g_beta, coef2, nullptr);
    const long n4 = rows * (C >> 2);
    hipLaunchKernelGGL(bn_cl_bwd_apply_kernel, dim3((unsigned)ceil_div(n4, (long)kBnclThreads)), dim3(kBnclThreads), 0, s, x, gy, coef,
                       coef2, relu, gx, n4, C);
    NRGBD_CHECK_LAUNCH();
    return NRGBD_OK;
}

/* bias + LeakyReLU on channels-last rows (autograd path of the R-Net): see the kernels above. */
extern "C" int nrgbd_bias_lrelu_cl_workgroups(long rows, int C) {
    if (rows <= 0 || C < 4 || C > 1024 || (C & 3)) return NRGBD_E_SHAPE;
    return nrgbd::bias_lrelu_groups(rows, C);
}

extern "C" int nrgbd_bias_lrelu_cl_fwd(const float* x, const float* bias, float slope, float* y, long rows, int C, void* stream) {
    using namespace nrgbd;
    if (!x || !bias || !y) return NRGBD_E_NULL;
    if (rows <= 0 || C < 4 || C > 1024 || (C & 3)) return NRGBD_E_SHAPE;
    const long n4 = rows * (C >> 2);
    hipLaunchKernelGGL(bias_lrelu_fwd_kernel, dim3((unsigned)ceil_div(n4, (long)kBnclThreads)), dim3(kBnclThreads), 0, (hipStream_t)stream,
                       x, bias, slope, y, n4, C);
    NRGBD_CHECK_LAUNCH();
    return NRGBD_OK;
}

extern "C" int nrgbd_bias_lrelu_cl_bwd(const float* y, const float* gy, float slope, float* gx, float* g_bias, float* partial,
                                       long rows, int C, void* stream) {
    using namespace nrgbd;
    if (!y || !gy || !gx || !g_bias || !partial) return NRGBD_E_NULL;
    if (rows <= 0 || C < 4 || C > 1024 || (C & 3)) return NRGBD_E_SHAPE;
    const int G = bias_lrelu_groups(rows, C);
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(bias_lrelu_bwd_kernel, dim3(G), dim3(kBnclThreads), 0, s, y, gy, slope, gx, partial, rows, C);
    hipLaunchKernelGGL(bias_lrelu_finalize_kernel, dim3((C + 15) / 16), dim3(kBnclThreads), 0, s, partial, G, C, g_bias);
    NRGBD_CHECK_LAUNCH();
    return NRGBD_OK;
}
