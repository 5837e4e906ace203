// softmax.hip — per-pixel reductions over the depth axis of a [D][n] volume (K4, K9, K13).
// HBM-bound: lanes run along the pixel axis (coalesced), each lane walks the D candidates of its
// pixel; for D <= 128 the column is held in registers so the volume is read once, written once.
#include "costvol.hpp"

namespace nrgbd {

// out = log_softmax_k(scale * a + b)   (models/basic.py:299-300, models/KVNET.py:172-173)
template <int DREG>  // DREG > 0: D <= DREG, column cached in registers; 0: three passes over memory
__global__ __launch_bounds__(256) void logsoftmax_d_kernel(const float* __restrict__ a,
                                                           const float* __restrict__ b, float scale,
                                                           float* __restrict__ out, int D, size_t n) {
    const size_t p = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (p >= n) return;
    auto load = [&](int k) -> float {
        float v = scale * a[(size_t)k * n + p];
        if (b) v = v + b[(size_t)k * n + p];
        return v;
    };
    if constexpr (DREG > 0) {
        float col[DREG];
        float m = -INFINITY;
#pragma unroll
        for (int k = 0; k < DREG; ++k) {
            col[k] = (k < D) ? load(k) : -INFINITY;
            m = fmaxf(m, col[k]);
        }
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < DREG; ++k)
            if (k < D) s += expf(col[k] - m);
        const float ls = logf(s);
#pragma unroll
        for (int k = 0; k < DREG; ++k)
            if (k < D) out[(size_t)k * n + p] = (col[k] - m) - ls;
    } else {
        float m = -INFINITY;
        for (int k = 0; k < D; ++k) m = fmaxf(m, load(k));
        float s = 0.f;
        for (int k = 0; k < D; ++k) s += expf(load(k) - m);
        const float ls = logf(s);
        for (int k = 0; k < D; ++k) out[(size_t)k * n + p] = (load(k) - m) - ls;
    }
}

// depth = sum_k exp(logp_k) * d_k, conf = max_k logp_k   (mutils/misc.py:532-548)
__global__ __launch_bounds__(256) void depth_regress_kernel(const float* __restrict__ logp,
                                                            const float* __restrict__ d_candi,
                                                            float* __restrict__ depth,
                                                            float* __restrict__ conf, int D, size_t n) {
    const size_t p = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (p >= n) return;
    float acc = 0.f, m = -INFINITY;
    for (int k = 0; k < D; ++k) {
        const float v = logp[(size_t)k * n + p];
        acc = acc + expf(v) * d_candi[k];
        m = fmaxf(m, v);
    }
    if (depth) depth[p] = acc;
    if (conf) conf[p] = m;
}

// The same reduction with FOUR threads per pixel (part p owns candidates p, p + 4, ...: up to 32 registers for D <= 128) and a
// workgroup of 64 pixels x 4 parts.  One thread per pixel leaves a 192x256 grid at 192 workgroups of 4 waves (less than one per
// CU) with 128 dependent-latency loads per thread: 57 us for 37.7 MB = 0.67 TB/s (rocprofv3, round 2).  Here every thread has
// 2 x 16 independent loads in flight and the grid is 768 workgroups.  Same arithmetic: the maximum is exact in any order and
// the sum of exponentials is taken as (p0 + p1) + (p2 + p3) of the four partial sums, each in candidate order.
template <int KMAX>   // candidates per part: D <= 4 * KMAX
__global__ __launch_bounds__(256) void logsoftmax_d4_kernel(const float* __restrict__ a, const float* __restrict__ b, float scale,
                                                            float* __restrict__ out, int D, size_t n) {
    __shared__ float red[2][4][64];
    const int pp = threadIdx.x & 63, part = threadIdx.x >> 6;
    const size_t p = (size_t)blockIdx.x * 64 + pp;
    const bool in = p < n;
    const size_t pc = in ? p : n - 1;
    float col[KMAX];
    float m = -INFINITY;
#pragma unroll
    for (int t = 0; t < KMAX; ++t) {
        const int k = part + 4 * t;
        float v = -INFINITY;
        if (k < D) {
            v = scale * a[(size_t)k * n + pc];
            if (b) v = v + b[(size_t)k * n + pc];
        }
        col[t] = v;
        m = fmaxf(m, v);
    }
    red[0][part][pp] = m;
    __syncthreads();
    m = fmaxf(fmaxf(red[0][0][pp], red[0][1][pp]), fmaxf(red[0][2][pp], red[0][3][pp]));
    float s = 0.f;
#pragma unroll
    for (int t = 0; t < KMAX; ++t)
        if (part + 4 * t < D) s += expf(col[t] - m);
    red[1][part][pp] = s;
    __syncthreads();
    s = (red[1][0][pp] + red[1][1][pp]) + (red[1][2][pp] + red[1][3][pp]);
    const float ls = logf(s);
    if (in) {
#pragma unroll
        for (int t = 0; t < KMAX; ++t) {
            const int k = part + 4 * t;
            if (k < D) out[(size_t)k * n + p] = (col[t] - m) - ls;
        }
    }
}

int launch_logsoftmax_d(const float* a, const float* b, float scale, float* out, int D, size_t n,
                        hipStream_t s) {
    if (D <= 64) {
        hipLaunchKernelGGL(logsoftmax_d4_kernel<16>, dim3(ceil_div((long)n, 64)), dim3(256), 0, s, a, b, scale, out, D, n);
    } else if (D <= 128) {
        hipLaunchKernelGGL(logsoftmax_d4_kernel<32>, dim3(ceil_div((long)n, 64)), dim3(256), 0, s, a, b, scale, out, D, n);
    } else {
        hipLaunchKernelGGL(logsoftmax_d_kernel<0>, dim3(ceil_div((long)n, 256)), dim3(256), 0, s, a, b, scale, out, D, n);
    }
    NRGBD_CHECK_LAUNCH();
    return NRGBD_OK;
}

// export epilogue (test_utils/export_res.py:43-75): expected depth, confidence exp(max_k logp) and the two uint16 maps of the
// .pgm files in ONE pass over the refined DPV: (map * scale) truncated toward zero, clamped to [0, 65535] (the reference's
// `.astype(np.uint16)` WRAPS beyond 65535: identical files for every depth * scale < 65536, saturation instead of a
// wrapped value beyond).  exp = exp_rn (common.hpp): the oracle's operation sequence, so the bytes are identical
__device__ __forceinline__ unsigned short to_u16(float v) {
    if (!(v > 0.f)) return 0;
    if (v >= 65535.f) return 65535;
    return (unsigned short)v;
}
__global__ __launch_bounds__(256) void export_depth_u16_kernel(const float* __restrict__ logp,
                                                               const float* __restrict__ d_candi, float depth_scale,
                                                               float conf_scale, float* __restrict__ depth,
                                                               float* __restrict__ conf, unsigned short* __restrict__ du,
                                                               unsigned short* __restrict__ cu, int D, size_t n) {
    const size_t p = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (p >= n) return;
    float acc = 0.f, m = -INFINITY;
    for (int k = 0; k < D; ++k) {
        const float v = logp[(size_t)k * n + p];
        acc = acc + exp_rn(v) * d_candi[k];
        m = fmaxf(m, v);
    }
    const float c = exp_rn(m);
    if (depth) depth[p] = acc;
    if (conf) conf[p] = c;
    if (du) du[p] = to_u16(acc * depth_scale);
    if (cu) cu[p] = to_u16(c * conf_scale);
}

}  // namespace nrgbd

namespace nrgbd {
// log_softmax over the channels of channels-last rows x [rows][C] (C = 64: 16 lanes x 16 bytes per row; C = 128: 32 lanes): the
// R-Net's last layer (models/Refine.py:104 F.log_softmax(conv2_2_out, dim=1)) after its convolution moved to the Winograd kernel,
// whose epilogue writes pixels channels-last.  One HBM pass (read + write, in place allowed); the reduction over a row is a
// butterfly of DPP / swizzle shuffles inside its lane group, in a fixed order.
template <int LPR>      // lanes per row
__global__ __launch_bounds__(256) void logsoftmax_rows_kernel(const float* __restrict__ x, float* __restrict__ y, long rows) {
    const long r = ((long)blockIdx.x * 256 + threadIdx.x) / LPR;
    const int q = threadIdx.x % LPR;
    const bool live = r < rows;
    const long rr = live ? r : rows - 1;
    const float4 v = *reinterpret_cast<const float4*>(x + (rr * LPR + q) * 4);
    float m = fmaxf(fmaxf(v.x, v.y), fmaxf(v.z, v.w));
#pragma unroll
    for (int o = LPR / 2; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    float s = (expf(v.x - m) + expf(v.y - m)) + (expf(v.z - m) + expf(v.w - m));
#pragma unroll
    for (int o = LPR / 2; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    const float ls = logf(s);
    if (live) *reinterpret_cast<float4*>(y + (r * LPR + q) * 4) = make_float4((v.x - m) - ls, (v.y - m) - ls, (v.z - m) - ls, (v.w - m) - ls);
}
}  // namespace nrgbd

namespace nrgbd {
// ---- training: backward of the two log-softmax forms, and the NLL loss of train_KVNet.py:103-120 in both layouts -------------
// out = log_softmax_k(scale * a + b):  g_z[k] = g[k] - exp(out[k]) * sum_k g[k],  g_a = scale * g_z,  g_b = g_z.
// Planar [D][n], four threads per pixel as logsoftmax_d4_kernel (sum of the four partial sums taken as (p0 + p1) + (p2 + p3)).
template <int KMAX>
__global__ __launch_bounds__(256) void logsoftmax_d4_bwd_kernel(const float* __restrict__ logp, const float* __restrict__ g, float scale,
                                                                float* __restrict__ gz, int D, size_t n) {
    __shared__ float red[4][64];
    const int pp = threadIdx.x & 63, part = threadIdx.x >> 6;
    const size_t p = (size_t)blockIdx.x * 64 + pp;
    const bool in = p < n;
    const size_t pc = in ? p : n - 1;
    float gv[KMAX], s = 0.f;
#pragma unroll
    for (int t = 0; t < KMAX; ++t) {
        const int k = part + 4 * t;
        gv[t] = k < D ? g[(size_t)k * n + pc] : 0.f;
        s += gv[t];
    }
    red[part][pp] = s;
    __syncthreads();
    s = (red[0][pp] + red[1][pp]) + (red[2][pp] + red[3][pp]);
    if (in) {
#pragma unroll
        for (int t = 0; t < KMAX; ++t) {
            const int k = part + 4 * t;
            if (k < D) gz[(size_t)k * n + p] = scale * (gv[t] - expf(logp[(size_t)k * n + p]) * s);
        }
    }
}
__global__ __launch_bounds__(256) void logsoftmax_d_bwd_kernel(const float* __restrict__ logp, const float* __restrict__ g, float scale,
                                                               float* __restrict__ gz, int D, size_t n) {
    const size_t p = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (p >= n) return;
    float s = 0.f;
    for (int k = 0; k < D; ++k) s += g[(size_t)k * n + p];
    for (int k = 0; k < D; ++k) gz[(size_t)k * n + p] = scale * (g[(size_t)k * n + p] - expf(logp[(size_t)k * n + p]) * s);
}
// channels-last rows [rows][C], lane group per row as logsoftmax_rows_kernel
template <int LPR>
__global__ __launch_bounds__(256) void logsoftmax_rows_bwd_kernel(const float* __restrict__ y, const float* __restrict__ g,
                                                                  float* __restrict__ gx, long rows) {
    const long r = ((long)blockIdx.x * 256 + threadIdx.x) / LPR;
    const int q = threadIdx.x % LPR;
    const bool live = r < rows;
    const long rr = live ? r : rows - 1;
    const float4 gv = *reinterpret_cast<const float4*>(g + (rr * LPR + q) * 4);
    const float4 yv = *reinterpret_cast<const float4*>(y + (rr * LPR + q) * 4);
    float s = (gv.x + gv.y) + (gv.z + gv.w);
#pragma unroll
    for (int o = LPR / 2; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if (live)
        *reinterpret_cast<float4*>(gx + (r * LPR + q) * 4) =
            make_float4(gv.x - expf(yv.x) * s, gv.y - expf(yv.y) * s, gv.z - expf(yv.z) * s, gv.w - expf(yv.w) * s);
}

// F.nll_loss(logp [1, D, h, w], target [1, h, w], ignore_index) with mean reduction (train_KVNet.py:103-120): element (k, p) of logp at
// k * sk + p * sp (planar: sk = n, sp = 1; channels-last: sk = 1, sp = D).  Pass 1: per-workgroup (sum of -logp[target], count of
// pixels that are not ignored) in a fixed-order tree; pass 2: one workgroup adds the partials in double: out = {sum / count, count}
// (0 / 0 = NaN when every pixel is ignored, as ATen).  A target outside [0, D) counts as ignored (ATen asserts).
__global__ __launch_bounds__(256) void nll_fwd_kernel(const float* __restrict__ logp, const long long* __restrict__ target, long long ignore,
                                                      int D, long n, long sk, long sp, float* __restrict__ partial) {
    __shared__ float sh[2][256];
    const int tid = threadIdx.x;
    const long p = (long)blockIdx.x * 256 + tid;
    float v = 0.f, c = 0.f;
    if (p < n) {
        const long long t = target[p];
        if (t != ignore && t >= 0 && t < D) { v = -logp[t * sk + p * sp]; c = 1.f; }
    }
    sh[0][tid] = v; sh[1][tid] = c;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (tid < o) { sh[0][tid] += sh[0][tid + o]; sh[1][tid] += sh[1][tid + o]; }
        __syncthreads();
    }
    if (tid == 0) { partial[2 * blockIdx.x] = sh[0][0]; partial[2 * blockIdx.x + 1] = sh[1][0]; }
}
__global__ __launch_bounds__(256) void nll_finalize_kernel(const float* __restrict__ partial, int G, float* __restrict__ out) {
    __shared__ double sh[2][256];
    const int tid = threadIdx.x;
    double v = 0.0, c = 0.0;
    for (int g = tid; g < G; g += 256) { v += (double)partial[2 * g]; c += (double)partial[2 * g + 1]; }
    sh[0][tid] = v; sh[1][tid] = c;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (tid < o) { sh[0][tid] += sh[0][tid + o]; sh[1][tid] += sh[1][tid + o]; }
        __syncthreads();
    }
    if (tid == 0) { out[0] = (float)(sh[0][0] / sh[1][0]); out[1] = (float)sh[1][0]; }
}
// g_logp[k, p] = -g_out / count at k = target[p] (not ignored), 0 elsewhere: the whole tensor in one pass, in logp's layout
__global__ __launch_bounds__(256) void nll_bwd_kernel(const long long* __restrict__ target, long long ignore, const float* __restrict__ gout,
                                                      const float* __restrict__ stat, float* __restrict__ glogp, int D, long n, int rows_layout) {
    const long e = (long)blockIdx.x * 256 + threadIdx.x;
    if (e >= (long)D * n) return;
    const long p = rows_layout ? e / D : e % n;
    const long long k = rows_layout ? e % D : e / n;
    const long long t = target[p];
    glogp[e] = (t == k && t != ignore) ? -gout[0] / stat[1] : 0.f;
}
}  // namespace nrgbd

extern "C" int nrgbd_logsoftmax_d_bwd(const float* logp, const float* g, float scale, float* gz, int D, long n, void* stream) {
    using namespace nrgbd;
    if (!logp || !g || !gz) return NRGBD_E_NULL;
    if (D <= 0 || n <= 0) return NRGBD_E_SHAPE;
    hipStream_t s = (hipStream_t)stream;
    if (D <= 64) hipLaunchKernelGGL(logsoftmax_d4_bwd_kernel<16>, dim3(ceil_div(n, 64)), dim3(256), 0, s, logp, g, scale, gz, D, (size_t)n);
    else if (D <= 128) hipLaunchKernelGGL(logsoftmax_d4_bwd_kernel<32>, dim3(ceil_div(n, 64)), dim3(256), 0, s, logp, g, scale, gz, D, (size_t)n);
    else hipLaunchKernelGGL(logsoftmax_d_bwd_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, s, logp, g, scale, gz, D, (size_t)n);
    NRGBD_CHECK_LAUNCH();
    return NRGBD_OK;
}

extern "C" int nrgbd_logsoftmax_rows_bwd(const float* y, const float* g, float* gx, long rows, int C, void* stream) {
    using namespace nrgbd;
    if (!y || !g || !gx) return NRGBD_E_NULL;
    if (rows <= 0 || (C != 64 && C != 128)) return NRGBD_E_SHAPE;
    const long threads = rows * (C / 4);
    const dim3 grid((unsigned)((threads + 255) / 256));
    if (C == 64) hipLaunchKernelGGL(logsoftmax_rows_bwd_kernel<16>, grid, dim3(256), 0, (hipStream_t)stream, y, g, gx, rows);
    else hipLaunchKernelGGL(logsoftmax_rows_bwd_kernel<32>, grid, dim3(256), 0, (hipStream_t)stream, y, g, gx, rows);
    NRGBD_CHECK_LAUNCH();
    return NRGBD_OK;
}

extern "C" int nrgbd_nll_workgroups(long n) { return n <= 0 ? 0 : (int)((n + 255) / 256); }

extern "C" int nrgbd_nll_fwd(const float* logp, const long long* target, long ignore_index, int D, long n, int channels_last,
                             float* partial, float* out, void* stream) {
    using namespace nrgbd;
    if (!logp || !target || !partial || !out) return NRGBD_E_NULL;
    if (D <= 0 || n <= 0) return NRGBD_E_SHAPE;
    const int G = nrgbd_nll_workgroups(n);
    hipLaunchKernelGGL(nll_fwd_kernel, dim3(G), dim3(256), 0, (hipStream_t)stream, logp, target, (long long)ignore_index, D, n,
                       channels_last ? 1L : n, channels_last ? (long)D : 1L, partial);
    hipLaunchKernelGGL(nll_finalize_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, partial, G, out);
    NRGBD_CHECK_LAUNCH();
    return NRGBD_OK;
}

extern "C" int nrgbd_nll_bwd(const long long* target, long ignore_index, const float* g_out, const float* stat, float* g_logp, int D,
                             long n, int channels_last, void* stream) {
    using namespace nrgbd;
    if (!target || !g_out || !stat || !g_logp) return NRGBD_E_NULL;
    if (D <= 0 || n <= 0) return NRGBD_E_SHAPE;
    hipLaunchKernelGGL(nll_bwd_kernel, dim3((unsigned)(((long)D * n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, target,
                       (long long)ignore_index, g_out, stat, g_logp, D, n, channels_last);
    NRGBD_CHECK_LAUNCH();
    return NRGBD_OK;
}

extern "C" int nrgbd_logsoftmax_rows(const float* x, float* y, long rows, int C, void* stream) {
    using namespace nrgbd;
    if (!x || !y) return NRGBD_E_NULL;
    if (rows <= 0 || (C != 64 && C != 128)) return NRGBD_E_SHAPE;
    const long threads = rows * (C / 4);
    const dim3 grid((unsigned)((threads + 255) / 256));
    if (C == 64) hipLaunchKernelGGL(logsoftmax_rows_kernel<16>, grid, dim3(256), 0, (hipStream_t)stream, x, y, rows);
    else hipLaunchKernelGGL(logsoftmax_rows_kernel<32>, grid, dim3(256), 0, (hipStream_t)stream, x, y, rows);
    NRGBD_CHECK_LAUNCH();
    return NRGBD_OK;
}

extern "C" int nrgbd_logsoftmax_d(const float* a, const float* b, float scale, float* out, int D,
                                  long n, void* stream) {
    if (!a || !out) return NRGBD_E_NULL;
    if (D <= 0 || n <= 0) return NRGBD_E_SHAPE;
    return nrgbd::launch_logsoftmax_d(a, b, scale, out, D, (size_t)n, (hipStream_t)stream);
}

extern "C" int nrgbd_depth_regress(const float* logp, const float* d_candi, float* depth,
                                   float* conf, int D, long n, void* stream) {
    using namespace nrgbd;
    if (!logp || !d_candi || (!depth && !conf)) return NRGBD_E_NULL;
    if (D <= 0 || n <= 0) return NRGBD_E_SHAPE;
    hipLaunchKernelGGL(depth_regress_kernel, dim3(ceil_div(n, 256)), dim3(256), 0,
                       (hipStream_t)stream, logp, d_candi, depth, conf, D, (size_t)n);
    NRGBD_CHECK_LAUNCH();
    return NRGBD_OK;
}

extern "C" int nrgbd_export_depth_u16(const float* logp, const float* d_candi, float depth_scale, float conf_scale,
                                      float* depth, float* conf, unsigned short* depth_u16, unsigned short* conf_u16,
                                      int D, long n, void* stream) {
    if (!logp || !d_candi) return NRGBD_E_NULL;
    if (!depth && !conf && !depth_u16 && !conf_u16) return NRGBD_E_NULL;
    if (D <= 0 || n <= 0) return NRGBD_E_SHAPE;
    hipLaunchKernelGGL(nrgbd::export_depth_u16_kernel, dim3(nrgbd::ceil_div(n, 256)), dim3(256), 0, (hipStream_t)stream,
                       logp, d_candi, depth_scale, conf_scale, depth, conf, depth_u16, conf_u16, D, (size_t)n);
    NRGBD_CHECK_LAUNCH();
    return NRGBD_OK;
}
