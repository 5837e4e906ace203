// pack.hip — NCHW (or channels-last) features (+ pooled RGB) -> NHWC texels.  HBM-bound: every input element is
// read once (coalesced along x) and every output texel written once as whole 16-B words.
//
// Replaces models/basic.py:254-263 (F.avg_pool2d + torch.cat) and produces the channel-last
// layout the sampling kernels need: one texel = Cp floats = Cp/4 x 16 B, so a bilinear tap is a
// run of aligned 16-byte loads.
#include "common.hpp"

namespace nrgbd {

constexpr int kPackTX = 64;  // pixels of one image row per workgroup

// grid: (ceil(w/64), h, N); block 256.  Dynamic LDS: Cp * (kPackTX + 1) floats.
__global__ __launch_bounds__(256) void pack_nhwc_kernel(const float* __restrict__ feat,
                                                        const float* __restrict__ rgb,
                                                        float* __restrict__ out, int Cf, int h,
                                                        int w, int pool, int Cp, int feat_nhwc,
                                                        float* __restrict__ rgb4) {
    extern __shared__ __attribute__((aligned(16))) float tile[];  // [Cp][kPackTX + 1]
    constexpr int LD = kPackTX + 1;
    const int x0 = blockIdx.x * kPackTX, y = blockIdx.y, n = blockIdx.z;
    const int npx = min(kPackTX, w - x0);
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const size_t hw = (size_t)h * w;

    // 1. CNN channels: one wave per channel row, lanes along x (256-B coalesced reads)
    if (!feat_nhwc) {
        const float* f = feat + ((size_t)n * Cf) * hw + (size_t)y * w + x0;
        for (int c = wv; c < Cf; c += 4) tile[c * LD + lane] = (lane < npx) ? f[(size_t)c * hw + lane] : 0.f;
    } else {  // features already channels-last (the matrix-core trunk): the row segment is one contiguous run
        const float* f = feat + (((size_t)n * h + y) * w + x0) * Cf;
        for (int t = tid; t < kPackTX * Cf; t += 256) {
            const int px = t / Cf, c = t - px * Cf;
            tile[c * LD + px] = (px < npx) ? f[t] : 0.f;
        }
    }

    // 2. pooled RGB channels Cf..Cf+2 (row-major window sum, then one division: avg_pool2d)
    const int n_rgb = rgb ? 3 : 0;
    if (rgb) {
        const int W = w * pool, H = h * pool;
        for (int t = tid; t < 3 * kPackTX; t += 256) {
            const int ch = t / kPackTX, x = t % kPackTX;
            float s = 0.f;
            if (x < npx) {
                const float* p = rgb + (((size_t)n * 3 + ch) * H + (size_t)y * pool) * W + (size_t)(x0 + x) * pool;
                for (int j = 0; j < pool; ++j)
                    for (int i = 0; i < pool; ++i) s += p[(size_t)j * W + i];
                s = s / (float)(pool * pool);
            }
            tile[(Cf + ch) * LD + x] = s;
        }
    }
    // 3. zero padding channels
    for (int t = tid; t < (Cp - Cf - n_rgb) * kPackTX; t += 256)
        tile[(Cf + n_rgb + t / kPackTX) * LD + (t % kPackTX)] = 0.f;
    __syncthreads();

    // 4. the tile's texels are one contiguous run of npx*Cp floats: linear 16-B stores
    const int cp4 = Cp >> 2;
    float4* o = reinterpret_cast<float4*>(out + (((size_t)n * h + y) * w + x0) * Cp);
    for (int i = tid; i < npx * cp4; i += 256) {
        const int px = i / cp4, c = (i - px * cp4) * 4;
        o[i] = make_float4(tile[(c + 0) * LD + px], tile[(c + 1) * LD + px],
                           tile[(c + 2) * LD + px], tile[(c + 3) * LD + px]);
    }
    // 5. the word that holds the pooled RGB (channels Cf .. Cf+3) once more as a compact [N][h][w][4] plane: the K-Net's warp
    // gathers it, and at the texels' 272-byte stride every tap of that kernel touched its own cache line
    if (rgb4) {
        float4* r4 = reinterpret_cast<float4*>(rgb4) + ((size_t)n * h + y) * w + x0;
        for (int px = tid; px < npx; px += 256)
            r4[px] = make_float4(tile[(Cf + 0) * LD + px], tile[(Cf + 1) * LD + px], tile[(Cf + 2) * LD + px], tile[(Cf + 3) * LD + px]);
    }
}

}  // namespace nrgbd

extern "C" int nrgbd_pack_nhwc(const float* feat, const float* rgb, float* out, int N, int Cf,
                               int h, int w, int pool, int Cp, int feat_channels_last, float* rgb4, void* stream) {
    if (!feat || !out) return NRGBD_E_NULL;
    if (rgb4 && (Cp < Cf + 4 || (reinterpret_cast<uintptr_t>(rgb4) & 15))) return NRGBD_E_ALIGN;
    if (N <= 0 || Cf <= 0 || h <= 0 || w <= 0 || pool <= 0 || h > 65535 || N > 65535) return NRGBD_E_SHAPE;
    if ((Cp & 3) || Cp < Cf + (rgb ? 3 : 0)) return NRGBD_E_ALIGN;
    if (reinterpret_cast<uintptr_t>(out) & 15) return NRGBD_E_ALIGN;
    const size_t lds = (size_t)Cp * (nrgbd::kPackTX + 1) * sizeof(float);
    if (lds > 160 * 1024) return NRGBD_E_SHAPE;
    dim3 grid(nrgbd::ceil_div(w, nrgbd::kPackTX), h, N);
    hipLaunchKernelGGL(nrgbd::pack_nhwc_kernel, grid, dim3(256), lds, (hipStream_t)stream, feat,
                       rgb, out, Cf, h, w, pool, Cp, feat_channels_last, rgb4);
    NRGBD_CHECK_LAUNCH();
    return NRGBD_OK;
}
