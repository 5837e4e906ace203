// conv3d_wino.hip — K-Net 3x3x3 convolution (64 -> 64, stride 1, pad 1, no bias) with the two in-plane dimensions in the
// Winograd domain F(2x2, 3x3), on the fp32 matrix cores of gfx950, BatchNorm3d work fused around it like conv3d.hip.
//
// Why: the ten 64->64 layers of the K-Net (models/basic.py:71-94) are 2/3 of a depth frame and conv3d.hip already runs them
// at 83-87 % of the fp32 MFMA peak — the only lever left is to issue fewer multiplies.  For every depth tap kd the 3x3
// in-plane convolution of a 2x2 output tile is  Y = A^T [ sum_ci sum_kd (G g_kd G^T) .* (B^T d_{z+kd-1} B) ] A :
// 16 multiplies per 4 outputs and (ci, kd) instead of 36, i.e. 2.25x fewer MFMAs, at the price of the two transforms (adds).
// Still exact-algorithm fp32 (no reduced precision anywhere); only the rounding order differs, as with any Winograd kernel
// (the reference's own cuDNN / MIOpen back ends use the same F(2,3) for these shapes).
//
// Decomposition.  Workgroup = 256 threads = one depth slice x 8 x 16 output pixels = 32 Winograd tiles x 64 output channels.
//   wave w  = output channels 16w .. 16w+15 for ALL 16 transform points xi and all 32 tiles:
//             v_mfma_f32_16x16x4_f32 with M = tiles (2 row tiles of 16), N = 16 channels, K = 16 input channels per
//             instruction group; 16 xi x 2 x f32x4 = 128 accumulator VGPRs.  Because a wave owns every xi of its
//             (tile, channel) pairs, the inverse transform A^T M A happens in registers: no exchange between waves.
//   K loop  = 4 input-channel blocks of 16 x 3 depth taps = 12 stages.  Per stage the (10 x 18)-pixel raw halo of one
//             input slice and channel block goes global -> registers (prefetched one stage ahead) -> BatchNorm / ReLU /
//             residual add -> LDS; each thread then transforms one (tile, 16-byte channel word, half of the xi rows):
//             12 LDS reads, 64 adds, 8 LDS writes of V[xi][tile][16 ch]; then 16 xi x (1 B load + 2 A reads + 8 MFMAs).
//   The transformed weights U[xi][kd][ci][co] = G g G^T are prepared once per weight update (host, cached) and packed so
//   that a wave's B operand of one (stage, xi) is one contiguous 1 KB line.
// LDS image of V: [xi*32 + tile][16 floats]; the 16-byte slot s of a tile is stored at slot (s + 2*((tile >> 3) & 1)) & 3,
// which makes every ds_read_b128 service group of the 16x16x4 A-operand pattern ({tiles 0-3, 12-15 at slot q} + {tiles 4-11
// at slot q+1}) hit 16 different bank quads.
#include "conv_tile.hpp"

namespace nrgbd {

constexpr int kWTH = 8, kWTW = 16;                  // output pixels per workgroup (one depth slice)
constexpr int kWHH = kWTH + 2, kWHW = kWTW + 2;     // raw halo
constexpr int kWRaw = kWHH * kWHW;                  // 180 pixels
constexpr int kWTiles = (kWTH / 2) * (kWTW / 2);    // 32 tiles: ty = tile >> 3 (0..3), tx = tile & 7
constexpr int kWCin = 64, kWCout = 64;
constexpr int kWStages = (kWCin / kCB) * 3;         // 12
constexpr int kWNPF = (kWRaw * 4 + 255) / 256;      // raw 16-byte words per thread per stage (3)

struct WinoArgs {
    const float* x;       // [D][H][W][64] raw input (pre-activation)
    const float* x_ss;    // [64][2] (scale, shift) applied to x, or null
    const float* res;     // [D][H][W][64] second operand added after activation, or null
    const float* res_ss;  // [64][2] for res, or null
    float* mat;           // [D][H][W][64] materialised input act(x) + act(res), or null
    const float* wp;      // packed Winograd-domain weights (12 stages x 16 xi x 4 waves x 64 lanes x 4)
    float* y;             // [D][H][W][64] raw convolution output
    float* stats;         // [workgroups][128]: per-channel sum and sum of squares of y, or null
    int x_relu, res_relu;
    int D, H, W;
};

typedef float f32x4w __attribute__((ext_vector_type(4)));

__device__ __forceinline__ int wino_slot(int xi, int tile, int slot) {
    return ((xi * kWTiles + tile) << 4) + (((slot + 2 * ((tile >> 3) & 1)) & 3) << 2);
}

template <bool RES>
__global__ __launch_bounds__(256, 2) void conv3d_wino_kernel(const WinoArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* raw = lds;                       // [180][16]
    float* V = lds + kWRaw * kCB;           // [16 xi][32 tiles][16]

    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int tiles_x = (a.W + kWTW - 1) / kWTW, tiles_y = (a.H + kWTH - 1) / kWTH;
    // XCD k (= blockIdx % 8) works on a contiguous eighth of the list; depth runs fastest, so the three workgroups that read
    // one input slice-tile (outputs z-1, z, z+1) are neighbours in the list and meet in the same L2
    int t = xcd_tile(blockIdx.x, gridDim.x, 1);
    const int tile_id = t;
    const int d = t % a.D; t /= a.D;
    const int tx0 = t % tiles_x, ty0 = t / tiles_x;
    const int x0 = tx0 * kWTW, y0 = ty0 * kWTH;
    (void)tiles_y;

    // ---- loader bookkeeping: word u of this thread = raw pixel (tid >> 2) + 64 u, 16-byte channel word tid & 3 ----
    const int c4 = tid & 3;
    unsigned pl_off[kWNPF];   // in-plane element offset of the pixel's channel c4*4
    unsigned pl_ok = 0, pl_own = 0;
#pragma unroll
    for (int u = 0; u < kWNPF; ++u) {
        const int px = (tid >> 2) + 64 * u;
        const int hy = px / kWHW, hx = px - hy * kWHW;
        const int gy = y0 + hy - 1, gx = x0 + hx - 1;
        const bool ok = px < kWRaw && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
        pl_off[u] = ok ? (unsigned)(((size_t)gy * a.W + gx) * kWCin + c4 * 4) : (unsigned)(c4 * 4);
        if (ok) pl_ok |= 1u << u;
        if (ok && hy >= 1 && hy <= kWTH && hx >= 1 && hx <= kWTW) pl_own |= 1u << u;
    }
    const unsigned plane = (unsigned)((size_t)a.H * a.W * kWCin);
    f32x4w pre[kWNPF], prer[RES ? kWNPF : 1];
    auto issue = [&](int s) {   // global loads of stage s (slice d + kd - 1, channel block cb) into pre / prer
        const int cb = s / 3, kd = s - 3 * cb;
        const int z = min(max(d + kd - 1, 0), a.D - 1);          // clamped: an outside slice is zeroed when published
        const unsigned base = (unsigned)z * plane + (unsigned)(cb * kCB);
#pragma unroll
        for (int u = 0; u < kWNPF; ++u) {
            pre[u] = *reinterpret_cast<const f32x4w*>(a.x + base + pl_off[u]);
            if constexpr (RES) prer[u] = *reinterpret_cast<const f32x4w*>(a.res + base + pl_off[u]);
        }
    };
    issue(0);

    // ---- this thread's transform item: (tile, channel word, half of the xi rows) ----
    const int ttile = tid >> 3, tword = (tid >> 1) & 3, thalf = tid & 1;
    const int tty = ttile >> 3, ttx = ttile & 7;
    const int rbase = ((2 * tty + thalf) * kWHW + 2 * ttx) * kCB + tword * 4;   // raw[(row)(col)][word] of the item's first pixel

    // ---- MFMA operand roles ----
    const int kq = lane >> 4, jj = lane & 15;
    f32x4w acc[16][2];
#pragma unroll
    for (int xi = 0; xi < 16; ++xi) { acc[xi][0] = f32x4w{0.f, 0.f, 0.f, 0.f}; acc[xi][1] = f32x4w{0.f, 0.f, 0.f, 0.f}; }
    const f32x4w* wbase = reinterpret_cast<const f32x4w*>(a.wp) + wv * 64 + lane;   // + (s*16 + xi) * 256

    for (int s = 0; s < kWStages; ++s) {
        const int cb = s / 3, kd = s - 3 * cb;
        const int z = d + kd - 1;
        const bool zin = z >= 0 && z < a.D;
        {   // (1) normalise / activate the prefetched raw words and publish them
            const int c = cb * kCB + c4 * 4;
            float ss[8] = {1.f, 0.f, 1.f, 0.f, 1.f, 0.f, 1.f, 0.f}, rs[8] = {1.f, 0.f, 1.f, 0.f, 1.f, 0.f, 1.f, 0.f};
            if (a.x_ss) {
#pragma unroll
                for (int e = 0; e < 8; ++e) ss[e] = a.x_ss[2 * c + e];
            }
            if (RES && a.res_ss) {
#pragma unroll
                for (int e = 0; e < 8; ++e) rs[e] = a.res_ss[2 * c + e];
            }
#pragma unroll
            for (int u = 0; u < kWNPF; ++u) {
                const int px = (tid >> 2) + 64 * u;
                if (px >= kWRaw) continue;
                f32x4w v = {0.f, 0.f, 0.f, 0.f};
                if (zin && ((pl_ok >> u) & 1u)) {   // zero padding applies to the ACTIVATED tensor
                    v = pre[u];
                    if (a.x_ss) {
                        v.x = __builtin_fmaf(v.x, ss[0], ss[1]); v.y = __builtin_fmaf(v.y, ss[2], ss[3]);
                        v.z = __builtin_fmaf(v.z, ss[4], ss[5]); v.w = __builtin_fmaf(v.w, ss[6], ss[7]);
                    }
                    if (a.x_relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
                    if constexpr (RES) {
                        f32x4w r = prer[u];
                        if (a.res_ss) {
                            r.x = __builtin_fmaf(r.x, rs[0], rs[1]); r.y = __builtin_fmaf(r.y, rs[2], rs[3]);
                            r.z = __builtin_fmaf(r.z, rs[4], rs[5]); r.w = __builtin_fmaf(r.w, rs[6], rs[7]);
                        }
                        if (a.res_relu) { r.x = fmaxf(r.x, 0.f); r.y = fmaxf(r.y, 0.f); r.z = fmaxf(r.z, 0.f); r.w = fmaxf(r.w, 0.f); }
                        v = v + r;
                    }
                    // the activated input is written once: by the workgroup that owns the voxel, at its centre tap
                    if (a.mat && kd == 1 && ((pl_own >> u) & 1u))
                        *reinterpret_cast<f32x4w*>(a.mat + (unsigned)z * plane + (unsigned)(cb * kCB) + pl_off[u]) = v;
                }
                *reinterpret_cast<f32x4w*>(raw + px * kCB + c4 * 4) = v;
            }
        }
        if (s + 1 < kWStages) issue(s + 1);   // (2) next stage's words fly while this stage transforms and multiplies
        __syncthreads();
        {   // (3) input transform B^T d B of this thread's (tile, word): rows first (2 of the 4 xi_y), then columns
            f32x4w r0[4], r1[4], r2[4];
#pragma unroll
            for (int cc = 0; cc < 4; ++cc) {
                r0[cc] = *reinterpret_cast<const f32x4w*>(raw + rbase + cc * kCB);
                r1[cc] = *reinterpret_cast<const f32x4w*>(raw + rbase + (kWHW + cc) * kCB);
                r2[cc] = *reinterpret_cast<const f32x4w*>(raw + rbase + (2 * kWHW + cc) * kCB);
            }
            // half 0: patch rows 0,1,2 -> xi_y 0 = r0 - r2, xi_y 1 = r1 + r2;  half 1: patch rows 1,2,3 -> xi_y 2 = r1' - r0' (= d2 - d1),
            // xi_y 3 = r0' - r2' (= d1 - d3), with r0', r1', r2' = this half's three rows
            f32x4w ya[4], yb[4];
#pragma unroll
            for (int cc = 0; cc < 4; ++cc) {
                ya[cc] = thalf ? (r1[cc] - r0[cc]) : (r0[cc] - r2[cc]);
                yb[cc] = thalf ? (r0[cc] - r2[cc]) : (r1[cc] + r2[cc]);
            }
            const int xa = (2 * thalf) * 4, xb = (2 * thalf + 1) * 4;
            *reinterpret_cast<f32x4w*>(V + wino_slot(xa + 0, ttile, tword)) = ya[0] - ya[2];
            *reinterpret_cast<f32x4w*>(V + wino_slot(xa + 1, ttile, tword)) = ya[1] + ya[2];
            *reinterpret_cast<f32x4w*>(V + wino_slot(xa + 2, ttile, tword)) = ya[2] - ya[1];
            *reinterpret_cast<f32x4w*>(V + wino_slot(xa + 3, ttile, tword)) = ya[1] - ya[3];
            *reinterpret_cast<f32x4w*>(V + wino_slot(xb + 0, ttile, tword)) = yb[0] - yb[2];
            *reinterpret_cast<f32x4w*>(V + wino_slot(xb + 1, ttile, tword)) = yb[1] + yb[2];
            *reinterpret_cast<f32x4w*>(V + wino_slot(xb + 2, ttile, tword)) = yb[2] - yb[1];
            *reinterpret_cast<f32x4w*>(V + wino_slot(xb + 3, ttile, tword)) = yb[1] - yb[3];
        }
        __syncthreads();
        {   // (4) 16 transform points x (B: one 1 KB line of U; A: two 16-tile row blocks of V) x 4 k-steps
            const f32x4w* wb = wbase + (size_t)s * 16 * 256;
            constexpr int BD = 3;
            f32x4w Bn[BD + 1], An[2][2];
#pragma unroll
            for (int b = 0; b < BD; ++b) Bn[b] = wb[b * 256];
            An[0][0] = *reinterpret_cast<const f32x4w*>(V + wino_slot(0, jj, kq));
            An[0][1] = *reinterpret_cast<const f32x4w*>(V + wino_slot(0, 16 + jj, kq));
#pragma unroll
            for (int xi = 0; xi < 16; ++xi) {
                const int cur = xi & 1, nxt = cur ^ 1;
                if (xi + 1 < 16) {
                    An[nxt][0] = *reinterpret_cast<const f32x4w*>(V + wino_slot(xi + 1, jj, kq));
                    An[nxt][1] = *reinterpret_cast<const f32x4w*>(V + wino_slot(xi + 1, 16 + jj, kq));
                }
                if (xi + BD < 16) Bn[(xi + BD) % (BD + 1)] = wb[(xi + BD) * 256];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    acc[xi][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(An[cur][0][e], Bn[xi % (BD + 1)][e], acc[xi][0], 0, 0, 0);
                    acc[xi][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(An[cur][1][e], Bn[xi % (BD + 1)][e], acc[xi][1], 0, 0, 0);
                }
            }
        }
        // the barrier of the next stage's step (1)->(3) also protects V: nobody rewrites it before every wave left (4)
    }

    // ---- inverse transform Y = A^T M A in registers + output + per-channel partial statistics ----
    // lane (kq, jj): output channel co = 16 wv + jj; register r of row block m = tile 16 m + 4 kq + r
    const int co = wv * 16 + jj;
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int m = 0; m < 2; ++m) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int tile = 16 * m + 4 * kq + r;
            const int ty = tile >> 3, tx = tile & 7;
            float tr[2][4];   // rows: t[a][xi_x] = sum_xi_y A^T[a][xi_y] M[xi_y][xi_x]
#pragma unroll
            for (int xx = 0; xx < 4; ++xx) {
                const float m0 = acc[0 + xx][m][r], m1 = acc[4 + xx][m][r], m2 = acc[8 + xx][m][r], m3 = acc[12 + xx][m][r];
                tr[0][xx] = (m0 + m1) + m2;
                tr[1][xx] = (m1 - m2) - m3;
            }
#pragma unroll
            for (int aa = 0; aa < 2; ++aa) {
                const float o0 = (tr[aa][0] + tr[aa][1]) + tr[aa][2];
                const float o1 = (tr[aa][1] - tr[aa][2]) - tr[aa][3];
                const int gy = y0 + 2 * ty + aa, gx = x0 + 2 * tx;
                if (gy < a.H) {
                    float* o = a.y + (((size_t)d * a.H + gy) * a.W + gx) * kWCout + co;
                    if (gx < a.W) { o[0] = o0; s1 += o0; s2 = __builtin_fmaf(o0, o0, s2); }
                    if (gx + 1 < a.W) { o[kWCout] = o1; s1 += o1; s2 = __builtin_fmaf(o1, o1, s2); }
                }
            }
        }
    }
    if (a.stats) {   // the wave owns its 16 channels: reduce over the 4 lanes (kq) that share a channel, no LDS needed
        s1 += __shfl_xor(s1, 16, 64); s2 += __shfl_xor(s2, 16, 64);
        s1 += __shfl_xor(s1, 32, 64); s2 += __shfl_xor(s2, 32, 64);
        if (kq == 0) {
            a.stats[(size_t)tile_id * 128 + co] = s1;
            a.stats[(size_t)tile_id * 128 + 64 + co] = s2;
        }
    }
}

}  // namespace nrgbd

extern "C" int nrgbd_conv3d_wino_workgroups(int D, int H, int W) {
    using namespace nrgbd;
    if (D <= 0 || H <= 0 || W <= 0) return NRGBD_E_SHAPE;
    return D * ceil_div(H, kWTH) * ceil_div(W, kWTW);
}

extern "C" int nrgbd_conv3d_wino_f32(const float* x, const float* x_ss, int x_relu, const float* res, const float* res_ss,
                                     int res_relu, float* materialized, const float* w_wino, float* y, float* stats, int D,
                                     int H, int W, void* stream) {
    using namespace nrgbd;
    if (!x || !w_wino || !y) return NRGBD_E_NULL;
    if (D <= 0 || H <= 0 || W <= 0) return NRGBD_E_SHAPE;
    if ((long)D * H * W * kWCin >= (1L << 32)) return NRGBD_E_SHAPE;   // 32-bit element offsets in the loader
    WinoArgs a{x, x_ss, res, res_ss, materialized, w_wino, y, stats, x_relu, res_relu, D, H, W};
    const int nwg = D * ceil_div(H, kWTH) * ceil_div(W, kWTW);
    const size_t lds = (size_t)(kWRaw * kCB + 16 * kWTiles * kCB) * sizeof(float);   // 11.5 KB raw + 32 KB V
    if (res) hipLaunchKernelGGL(conv3d_wino_kernel<true>, dim3(nwg), dim3(256), lds, (hipStream_t)stream, a);
    else     hipLaunchKernelGGL(conv3d_wino_kernel<false>, dim3(nwg), dim3(256), lds, (hipStream_t)stream, a);
    NRGBD_CHECK_LAUNCH();
    return NRGBD_OK;
}
