// glue.hip — the small data-movement passes between the matrix-core kernels of the inference frame that used to be ATen launches
// (round 6: profiles/r5_frame_B_dispatches.txt listed 0.31 ms of avg_pool2d / copy / cat per frame at config B):
//   nrgbd_avgpool_cl        k x k / stride-k average pooling of a channels-last map (the SPP windows, psm_submodule.py:100-117)
//   nrgbd_scatter_channels  a strided [C][H][W] view -> channels coff .. coff+C-1 of the pixels of `n_rep` channels-last images
//                           (the image features the R-Net concatenates behind its candidate channels, Refine.py:88-98)
#include "common.hpp"

namespace nrgbd {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// thread = (output pixel, 16-byte channel word); the k x k window is summed row-major (avg_pool2d's order), one division.
// Floor semantics: the ragged border of a map whose sides are not multiples of k is dropped, as avg_pool2d does.
__global__ __launch_bounds__(256) void avgpool_cl_kernel(const float* __restrict__ x, float* __restrict__ y, int N, int H, int W,
                                                         int C4, int k) {
    const int Ho = H / k, Wo = W / k;
    const long total = (long)N * Ho * Wo * C4;
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int c4 = (int)(idx % C4);
    long p = idx / C4;
    const int xo = (int)(p % Wo); p /= Wo;
    const int yo = (int)(p % Ho);
    const int n = (int)(p / Ho);
    const f32x4* src = reinterpret_cast<const f32x4*>(x) + (((size_t)n * H + (size_t)yo * k) * W + (size_t)xo * k) * C4 + c4;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    for (int j = 0; j < k; ++j) {
        const f32x4* row = src + (size_t)j * W * C4;
        for (int i = 0; i < k; ++i) s = s + row[(size_t)i * C4];
    }
    const float d = (float)(k * k);
    reinterpret_cast<f32x4*>(y)[idx] = f32x4{s.x / d, s.y / d, s.z / d, s.w / d};
}

// 8 x 8 windows of a large map (the SPP's finest level on the deep map: 126 MB at config B): a 256-thread block = (8 window rows) x
// (32 channel words) of ONE output pixel of a 128-channel map (generally: 256 / C4 rows at a time); every thread sums its row's 8
// words from 8 independent loads, the rows meet in LDS in row order.  One thread per output word walked its 64 words one after the
// other: 43 us at config B against 20 us of HBM time.
__global__ __launch_bounds__(256) void avgpool8_cl_kernel(const float* __restrict__ x, float* __restrict__ y, int H, int W, int C4) {
    __shared__ f32x4 part[256];
    const int rows = 256 / C4;                          // window rows in flight (8 for C = 128, more for narrower maps: capped at 8)
    const int tid = threadIdx.x, c4 = tid % C4, j = tid / C4;
    const int Wo = W >> 3, Ho = H >> 3;
    const int xo = blockIdx.x % Wo, yo = (blockIdx.x / Wo) % Ho, n = blockIdx.x / (Wo * Ho);
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    if (j < 8 && j < rows) {
        for (int jj = j; jj < 8; jj += (rows < 8 ? rows : 8)) {
            const f32x4* row = reinterpret_cast<const f32x4*>(x) + (((size_t)n * H + (size_t)yo * 8 + jj) * W + (size_t)xo * 8) * C4 + c4;
            f32x4 v[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = row[(size_t)i * C4];
#pragma unroll
            for (int i = 0; i < 8; ++i) s = s + v[i];
        }
    }
    part[tid] = s;
    __syncthreads();
    if (j == 0) {
        const int nr = rows < 8 ? rows : 8;
        for (int r = 1; r < nr; ++r) s = s + part[r * C4 + c4];
        reinterpret_cast<f32x4*>(y)[(size_t)blockIdx.x * C4 + c4] = f32x4{s.x / 64.f, s.y / 64.f, s.z / 64.f, s.w / 64.f};
    }
}

// thread = (pixel, channel); dst[rep][pixel * ldy + coff + c] = src[c * sc + y * sy + x * sx]
__global__ __launch_bounds__(256) void scatter_channels_kernel(const float* __restrict__ src, long sc, long sy, long sx, int C, int H,
                                                               int W, float* __restrict__ dst, int ldy, int coff, int n_rep,
                                                               long rep_stride) {
    const long total = (long)H * W * C;
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int c = (int)(idx % C);
    const long p = idx / C;
    const int xx = (int)(p % W), yy = (int)(p / W);
    const float v = src[(size_t)c * sc + (size_t)yy * sy + (size_t)xx * sx];
    float* o = dst + (size_t)p * ldy + coff + c;
    for (int r = 0; r < n_rep; ++r) o[(size_t)r * rep_stride] = v;
}

// the same for a source whose channels are contiguous (sc == 1: a channels-last map), 16-byte words: thread = (pixel, word)
__global__ __launch_bounds__(256) void scatter_channels_cl4_kernel(const float* __restrict__ src, long sy, long sx, int C4, int H, int W,
                                                                   float* __restrict__ dst, int ldy, int coff, int n_rep, long rep_stride) {
    const long total = (long)H * W * C4;
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int c4 = (int)(idx % C4);
    const long p = idx / C4;
    const int xx = (int)(p % W), yy = (int)(p / W);
    const f32x4 v = *reinterpret_cast<const f32x4*>(src + (size_t)yy * sy + (size_t)xx * sx + 4 * c4);
    float* o = dst + (size_t)p * ldy + coff + 4 * c4;
    for (int r = 0; r < n_rep; ++r) *reinterpret_cast<f32x4*>(o + (size_t)r * rep_stride) = v;
}

// ... and for a few planar channels (C <= 4: the RGB image): thread = pixel, lanes along x read every plane coalesced
__global__ __launch_bounds__(256) void scatter_channels_planar_kernel(const float* __restrict__ src, long sc, long sy, int C, int H, int W,
                                                                      float* __restrict__ dst, int ldy, int coff, int n_rep, long rep_stride) {
    const long total = (long)H * W;
    const long p = (long)blockIdx.x * 256 + threadIdx.x;
    if (p >= total) return;
    const int xx = (int)(p % W), yy = (int)(p / W);
    float v[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) v[c] = c < C ? src[(size_t)c * sc + (size_t)yy * sy + xx] : 0.f;
    for (int r = 0; r < n_rep; ++r) {
        float* o = dst + (size_t)r * rep_stride + (size_t)p * ldy + coff;
#pragma unroll
        for (int c = 0; c < 4; ++c)
            if (c < C) o[c] = v[c];
    }
}

}  // namespace nrgbd

extern "C" int nrgbd_avgpool_cl(const float* x, float* y, int N, int H, int W, int C, int k, void* stream) {
    using namespace nrgbd;
    if (!x || !y) return NRGBD_E_NULL;
    if (N <= 0 || H <= 0 || W <= 0 || C < 4 || (C & 3) || k <= 0 || H / k <= 0 || W / k <= 0) return NRGBD_E_SHAPE;
    if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y)) & 15) return NRGBD_E_ALIGN;
    const long total = (long)N * (H / k) * (W / k) * (C >> 2);
    const int C4 = C >> 2;
    if (k == 8 && C4 <= 256 && 256 % C4 == 0 && total >= 4096 && total / C4 < (1L << 31))
        hipLaunchKernelGGL(avgpool8_cl_kernel, dim3((unsigned)(total / C4)), dim3(256), 0, (hipStream_t)stream, x, y, H, W, C4);
    else
        hipLaunchKernelGGL(avgpool_cl_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, y, N, H, W,
                           C4, k);
    NRGBD_CHECK_LAUNCH();
    return NRGBD_OK;
}

extern "C" int nrgbd_scatter_channels(const float* src, long stride_c, long stride_y, long stride_x, int C, int H, int W, float* dst,
                                      int ldy, int coff, int n_rep, long rep_stride, void* stream) {
    using namespace nrgbd;
    if (!src || !dst) return NRGBD_E_NULL;
    if (C <= 0 || H <= 0 || W <= 0 || ldy < coff + C || coff < 0 || n_rep <= 0) return NRGBD_E_SHAPE;
    hipStream_t st = (hipStream_t)stream;
    const bool al16 = ((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst)) & 15) == 0;
    if (stride_c == 1 && (C & 3) == 0 && al16 && !((stride_y | stride_x | ldy | coff | rep_stride) & 3)) {
        const long total = (long)H * W * (C >> 2);
        hipLaunchKernelGGL(scatter_channels_cl4_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, src, stride_y, stride_x,
                           C >> 2, H, W, dst, ldy, coff, n_rep, rep_stride);
    } else if (C <= 4 && stride_x == 1) {
        const long total = (long)H * W;
        hipLaunchKernelGGL(scatter_channels_planar_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, src, stride_c, stride_y,
                           C, H, W, dst, ldy, coff, n_rep, rep_stride);
    } else {
        const long total = (long)H * W * C;
        hipLaunchKernelGGL(scatter_channels_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, src, stride_c, stride_y,
                           stride_x, C, H, W, dst, ldy, coff, n_rep, rep_stride);
    }
    NRGBD_CHECK_LAUNCH();
    return NRGBD_OK;
}
