// conv_few.hip — 3x3 convolution (stride 1, pad 1, + bias, + LeakyReLU) with a HANDFUL of output channels, on the vector ALUs.
//
// Why.  The R-Net's full-resolution layer conv2 (models/Refine.py:64-66: conv2d_leakyRelu(D + 3, D + 3)) has 67 = 64 + 3 output
// channels.  The Winograd kernel (wino_pc.hip) works in groups of 64 columns; the 3 columns beyond the group ran as a 32-column pass
// on its HALF form: a whole second input transform of the 2 x 768 x 1024 x 80 buffer for three columns — 0.38 ms per frame at config
// B (profiles/r5_frame_B_dispatches.txt), the single most wasteful launch of the frame.  Three outputs per pixel are 67 x 9 x 3 = 1,809
// multiply-adds: 0.07 ms of plain v_fmac_f32 for the whole batch.  No matrix core is needed, only the input tile in LDS.
//
//   workgroup = 16 x 16 output pixels (256 threads, one pixel each) of one image, all CO <= 4 output channels;
//   input     = channel blocks of 16 of the 18 x 18 halo in LDS as [pixel][20 floats] (pitch 20: eight consecutive pixels' 16-byte
//               words fall on eight different bank quads): ONE 26 KB buffer — six workgroups per CU hide the scalar-load and LDS
//               latencies of each other (a second buffer halves the occupancy: 0.185 vs 0.13 ms at config B); block c + 1 is
//               fetched into registers while block c is multiplied and published between two barriers;
//   weights   = [block][tap][co][16] floats: uniform per wave, so they arrive through the scalar cache (s_load_dwordx16) and enter
//               the v_fmac_f32 as SGPR operands — no vector register, no LDS;
//   summation = per output: bias first, then (channel block, tap, channel) ascending: one fp32 FMA chain.
#include "common.hpp"

namespace nrgbd {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kFewT = 16, kFewHS = kFewT + 2, kFewHalo = kFewHS * kFewHS;   // 324 halo pixels
constexpr int kFewPitch = 20;                                                // floats per halo pixel in LDS (16 + 4 pad)
constexpr int kFewNPF = (kFewHalo * 4 + 255) / 256;                          // 16-byte words per thread and channel block (6)

struct ConvFewArgs {
    const float* x;     // [N][H][W][ldx], channels 0 .. Cin-1 of a pixel are read (Cin % 16 == 0; padding channels must be zero or carry zero weights)
    const float* wp;    // [Cin/16][9][CO][16]
    const float* bias;  // [CO] or null
    float* y;           // [N][H][W][ldy]; output column co goes to y[pixel * ldy + ycoff + co]
    int N, H, W, Cin, ldx, ldy, ycoff, lrelu;
};

template <int CO>
__global__ __launch_bounds__(256, 6) void conv_few_kernel(const ConvFewArgs a) {
    __shared__ __attribute__((aligned(16))) float tile[1][kFewHalo * kFewPitch];
    const int tid = threadIdx.x, px = tid & 15, py = tid >> 4;
    const int tiles_x = (a.W + kFewT - 1) / kFewT, tiles_y = (a.H + kFewT - 1) / kFewT;
    int t = blockIdx.x;
    const int tx = t % tiles_x; t /= tiles_x;
    const int ty = t % tiles_y; const int n = t / tiles_y;
    const int x0 = tx * kFewT, y0 = ty * kFewT;
    const int nblk = a.Cin >> 4;

    // this thread's words of a channel block: word u = halo pixel (tid >> 2) + 64 u, 16-byte word tid & 3
    const int c4 = tid & 3;
    unsigned off[kFewNPF];
    unsigned ok = 0;
#pragma unroll
    for (int u = 0; u < kFewNPF; ++u) {
        const int hv = (tid >> 2) + 64 * u;
        const int hy = hv / kFewHS, hx = hv - hy * kFewHS;
        const int gy = y0 + hy - 1, gx = x0 + hx - 1;
        const bool in = hv < kFewHalo && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
        off[u] = in ? (unsigned)((((size_t)n * a.H + gy) * a.W + gx) * a.ldx + c4 * 4) : (unsigned)(c4 * 4);
        if (in) ok |= 1u << u;
    }
    f32x4 pre[kFewNPF];
#pragma unroll
    for (int u = 0; u < kFewNPF; ++u) pre[u] = *reinterpret_cast<const f32x4*>(a.x + off[u]);
    auto publish = [&](int buf) {
#pragma unroll
        for (int u = 0; u < kFewNPF; ++u) {
            const int hv = (tid >> 2) + 64 * u;
            if (hv < kFewHalo)
                *reinterpret_cast<f32x4*>(&tile[buf][hv * kFewPitch + c4 * 4]) = ((ok >> u) & 1u) ? pre[u] : f32x4{0.f, 0.f, 0.f, 0.f};
        }
    };
    publish(0);
    __syncthreads();

    float acc[CO];
#pragma unroll
    for (int co = 0; co < CO; ++co) acc[co] = a.bias ? a.bias[co] : 0.f;

    for (int c = 0; c < nblk; ++c) {
        constexpr int buf = 0;
        if (c + 1 < nblk) {
#pragma unroll
            for (int u = 0; u < kFewNPF; ++u) pre[u] = *reinterpret_cast<const f32x4*>(a.x + off[u] + (c + 1) * 16);
        }
        const float* w = a.wp + (size_t)c * (9 * CO * 16);     // wave-uniform: scalar loads
        // one tap row at a time (not unrolled): with all nine taps in flight the 36 operand words and 432 scalar weights of a channel
        // block overflow both register files
#pragma unroll 1
        for (int ky = 0; ky < 3; ++ky) {
            const float* trow = &tile[buf][((py + ky) * kFewHS + px) * kFewPitch];
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const float* tp = trow + kx * kFewPitch;
                f32x4 v[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) v[q] = *reinterpret_cast<const f32x4*>(tp + 4 * q);
#pragma unroll
                for (int co = 0; co < CO; ++co) {
                    const float* wr = w + ((ky * 3 + kx) * CO + co) * 16;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        acc[co] = __builtin_fmaf(v[q].x, wr[4 * q + 0], acc[co]);
                        acc[co] = __builtin_fmaf(v[q].y, wr[4 * q + 1], acc[co]);
                        acc[co] = __builtin_fmaf(v[q].z, wr[4 * q + 2], acc[co]);
                        acc[co] = __builtin_fmaf(v[q].w, wr[4 * q + 3], acc[co]);
                    }
                }
            }
        }
        __syncthreads();                          // every wave has read block c: the one buffer may take block c + 1
        if (c + 1 < nblk) publish(0);
        __syncthreads();
    }
    const int gy = y0 + py, gx = x0 + px;
    if (gy < a.H && gx < a.W) {
        float* o = a.y + (((size_t)n * a.H + gy) * a.W + gx) * a.ldy + a.ycoff;
#pragma unroll
        for (int co = 0; co < CO; ++co) {
            float z = acc[co];
            if (a.lrelu) z = z > 0.f ? z : 0.01f * z;
            o[co] = z;
        }
    }
}

}  // namespace nrgbd

extern "C" int nrgbd_conv2d_few_f32(const float* x, int ldx, const float* w_packed, const float* bias, int out_lrelu, float* y, int ldy,
                                    int ycoff, int N, int H, int W, int Cin, int Cout, void* stream) {
    using namespace nrgbd;
    if (!x || !w_packed || !y) return NRGBD_E_NULL;
    if (N <= 0 || H <= 0 || W <= 0 || Cin <= 0 || (Cin & 15) || ldx < Cin || (ldx & 3) || Cout < 1 || Cout > 4 || ycoff < 0 ||
        ldy < ycoff + Cout)
        return NRGBD_E_SHAPE;
    if (reinterpret_cast<uintptr_t>(x) & 15) return NRGBD_E_ALIGN;
    if ((long)N * H * W * ldx >= (1L << 32)) return NRGBD_E_SHAPE;      // 32-bit element offsets in the loader
    ConvFewArgs a{x, w_packed, bias, y, N, H, W, Cin, ldx, ldy, ycoff, out_lrelu};
    const int nwg = ceil_div(W, kFewT) * ceil_div(H, kFewT) * N;
    hipStream_t st = (hipStream_t)stream;
    switch (Cout) {
        case 1: hipLaunchKernelGGL(conv_few_kernel<1>, dim3(nwg), dim3(256), 0, st, a); break;
        case 2: hipLaunchKernelGGL(conv_few_kernel<2>, dim3(nwg), dim3(256), 0, st, a); break;
        case 3: hipLaunchKernelGGL(conv_few_kernel<3>, dim3(nwg), dim3(256), 0, st, a); break;
        default: hipLaunchKernelGGL(conv_few_kernel<4>, dim3(nwg), dim3(256), 0, st, a); break;
    }
    NRGBD_CHECK_LAUNCH();
    return NRGBD_OK;
}
