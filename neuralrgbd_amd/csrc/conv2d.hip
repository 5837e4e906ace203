// conv2d.hip — 3x3 convolutions of the 2-D feature CNN (stride 1, pad = dilation) on the fp32 matrix cores of
// gfx950, with the BatchNorm2d work of the reference fused around them, plus the two small channels-last helpers
// the trunk needs around the few layers that stay on the vendor library.
//
// Replaces, per trunk layer (models/psm_submodule.py:10-16 convbn, :31-50 BasicBlock, :90-167 feature_extraction):
//   nn.Conv2d(Cin, Cout, 3, stride 1, padding = dilation, dilation, bias=False) -> implicit GEMM on v_mfma_f32_32x32x2_f32
//   nn.BatchNorm2d with batch statistics (the reference never leaves train() mode)
//        statistics -> per-workgroup (sum, sum of squares) partials from the conv epilogue (bn_finalize reduces in fp64)
//        normalise + affine + ReLU + residual add -> applied by the NEXT layer while it loads its input tile
// i.e. the same one-pass-per-layer scheme as the K-Net (conv3d.hip), on [N][H][W][C] channels-last activations.
// The reference runs conv, BN statistics, BN apply, ReLU and the residual add as five passes per layer.
// The optional (bias, LeakyReLU 0.01) epilogue is the R-Net's conv form (models/m_submodule.py:18-27).
//
// Tiling: workgroup = 256 threads = 4 waves = 16 x 16 output pixels x COUT channels of one image; a wave owns
// 4 rows x 16 x = two 32-row MFMA tiles x COUT/32 column tiles.  K loop over channel blocks of 16: the
// (16+2d)^2 halo tile of the block sits in LDS as [pixel][16] with XOR-swizzled 16-B slots (conv_tile.hpp),
// then 9 taps x 2 k-groups, each = 2 A reads (LDS, b128) + COUT/32 B loads (L2, one 1 KB line per wave) +
// 8 * COUT/32 MFMAs.  The raw words of block c+1 (input and residual operand) are fetched into registers
// while the MFMAs of block c run, one 16-B load per step, and normalised / published to LDS after the loop.
#include <cstdlib>

#include "conv_tile.hpp"

namespace nrgbd {

constexpr int kT2 = 16;  // output tile edge (pixels)

struct Conv2dArgs {
    const float* x;       // [N][H][W][Cin] raw input (pre-activation)
    const float* x_ss;    // [Cin][2] (scale, shift) applied to x, or null = identity
    const float* res;     // [N][H][W][Cin] second operand added after activation, or null
    const float* res_ss;  // [Cin][2] for res, or null = identity
    float* mat;           // [N][H][W][Cin]: materialised input act(x) + act(res), or null
    const float* wp;      // packed weights (conv_pack_weights, 9 taps)
    const float* bias;    // [Cout] added in the epilogue, or null
    float* y;             // [N][H][W][Cout] convolution output
    float* stats;         // [num_workgroups][2*Cout]: per-channel sum and sum of squares of y, or null
    int x_relu, res_relu, out_lrelu;
    int N, H, W, Cin;
    int xcd;              // re-map workgroups so each XCD owns a contiguous run of tiles (conv_tile.hpp)
    // generalised output addressing (R-Net, models/Refine.py:51-107): defaults = plain [N][H][W][COUT]
    int ldy;              // pixel stride of y in floats (>= ycoff + cout_valid): lets a layer write straight INTO a concat buffer
    int ycoff;            // first channel of y this layer writes
    int cout_valid;       // columns that exist (< COUT when Cout was padded to the 32-wide fragments)
    int up;               // 1: this launch is sub-pixel phase (pa, pb) of a stride-2 transposed conv: 2x2 taps, output pixel
    int pa, pb;           //    (2y+pa, 2x+pb) of a [N][2H][2W] tensor
    float* planar;        // EPI = 1: log_softmax over the COUT channels, written as [N][COUT][H][W] (the R-Net's last layer)
    int xs;               // NTAP = 1 only: spatial stride of the input (0 / 1 = none): x is [N][H*xs][W*xs][Cin] and output pixel (y, x)
                          // reads input pixel (y*xs, x*xs) — the stride-2 1x1 shortcut of psm_submodule.py:127-131 without a gather pass
};

// NTAP = 9: 3x3 convolution.  NTAP = 4: one sub-pixel phase of ConvTranspose2d(k=4, s=2, p=1) (m_submodule.py:37-45): output
// pixel (2y+pa, 2x+pb) sums the 2x2 input neighbourhood rows {y-1+pa, y+pa} x cols {x-1+pb, x+pb} — a 2x2-tap convolution
// on the same halo tile; the four phases are four launches, so no multiply-by-zero is ever issued.
// EPI = 1: bias, then log_softmax over the COUT columns of every pixel, stored planar (Refine.py:101-105).
template <int COUT, int DIL, bool RES, int NTAP = 9, int EPI = 0>
__global__ __launch_bounds__(256, COUT <= 32 ? 3 : 2) void conv2d_mfma_kernel(const Conv2dArgs a) {
    constexpr int HS = kT2 + 2 * DIL, HALO = HS * HS;   // halo tile edge / pixels (324 | 400)
    constexpr int NPF = (HALO * (kCB / 4) + 255) / 256;  // 16-B words per thread per channel block (6 | 7)
    constexpr int NF = COUT / 32;                        // 32-column output fragments
    constexpr int G4 = kCB / 8;                          // k-groups per block (4 k-steps = 8 channels each)
    constexpr int NSTEP = NTAP * G4;
    static_assert(NTAP == 9 || ((NTAP == 4 || NTAP == 1) && DIL == 1), "2x2 taps: transposed-conv phases / stride-2 convs on a "
                                                                         "space-to-depth input; 1 tap: 1x1 convolutions");
    // the next block's words are fetched one per step where the step loop is long enough, otherwise all in step 0 (1)
    constexpr bool PF_SPREAD = 2 + (RES ? 2 : 1) * NPF <= NSTEP;
    static_assert(PF_SPREAD || NSTEP >= 2, "prefetch does not fit the step loop");
    extern __shared__ __attribute__((aligned(16))) float lds[];  // [HALO][kSV]; reused for the statistics

    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int tiles_x = (a.W + kT2 - 1) / kT2, tiles_y = (a.H + kT2 - 1) / kT2;
    int t = xcd_tile(blockIdx.x, gridDim.x, a.xcd);
    const int tile_id = t;
    const int tx = t % tiles_x; t /= tiles_x;
    const int ty = t % tiles_y; const int n = t / tiles_y;
    const int x0 = tx * kT2, y0 = ty * kT2;
    const int Cin = a.Cin, ncblk = Cin / kCB;

    // this lane's A rows: wave -> 4-row band; tile m -> 2 rows of the band
    const int i = lane & 31, khalf = lane >> 5;
    int dy, px;
    row_to_yx(i, dy, px);
    const int wy = wv * 4;
    // m = 0, tap (0,0); m = 1 is two rows further.  A transposed-conv phase starts its 2x2 window at halo offset (pa, pb).
    // a 1x1 convolution reads the centre of the (unused) one-pixel halo
    // up == 2: all four sub-pixel phases of a transposed convolution in ONE launch (blockIdx.y = phase, its weights the
    // phase's quarter of w_packed): four times the workgroups of a per-phase launch, which at the R-Net's grids fill half the chip
    int pa = a.pa, pb = a.pb;
    const float* wpk = a.wp;
    if constexpr (NTAP == 4) {
        if (a.up == 2) { pa = blockIdx.y >> 1; pb = blockIdx.y & 1; wpk += (size_t)blockIdx.y * 4 * a.Cin * COUT; }
    }
    const int hv0 = (wy + dy) * HS + px + (NTAP == 4 ? pa * HS + pb : NTAP == 1 ? HS + 1 : 0);

    f32x16 acc[2][NF];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int f = 0; f < NF; ++f)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[m][f][e] = 0.f;

    // prefetch bookkeeping: word u of this thread = halo pixel (tid>>2) + 64u, 16-B word tid&3 of the block
    const int c4 = tid & 3;
    unsigned pf_off[NPF];   // element offset of the pixel's channel c4*4 (a harmless in-tensor offset when outside)
    unsigned pf_ok = 0;     // bit u: the pixel is inside the image (outside = zero padding)
    unsigned pf_own = 0;    // bit u: the pixel belongs to this tile's interior (materialise target)
    f32x4 pre[NPF], prer[RES ? NPF : 1];
#pragma unroll
    for (int u = 0; u < NPF; ++u) {
        const int hv = (tid >> 2) + 64 * u;
        const int hy = hv / HS, hx = hv - hy * HS;
        const int gy = y0 + hy - DIL, gx = x0 + hx - DIL;
        const bool ok = hv < HALO && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
        const int xs = (NTAP == 1 && a.xs > 1) ? a.xs : 1;
        pf_off[u] = ok ? (unsigned)((((size_t)n * (a.H * xs) + (size_t)gy * xs) * (a.W * xs) + (size_t)gx * xs) * Cin + c4 * 4) : (unsigned)(c4 * 4);
        if (ok) pf_ok |= 1u << u;
        if (ok && hy >= DIL && hy < DIL + kT2 && hx >= DIL && hx < DIL + kT2) pf_own |= 1u << u;
    }
#pragma unroll
    for (int u = 0; u < NPF; ++u) {
        // unconditional loads (lanes outside the image read a valid dummy word that is zeroed when published):
        // predicated loads into a register array end up in scratch
        pre[u] = *reinterpret_cast<const f32x4*>(a.x + pf_off[u]);
        if constexpr (RES) prer[u] = *reinterpret_cast<const f32x4*>(a.res + pf_off[u]);
    }

    const int wstep = ncblk * (G4 * NF * 64);  // f32x4 per tap
    for (int cblk = 0; cblk < ncblk; ++cblk) {
        {   // normalise / activate the prefetched words of this block and publish them to LDS
            const int c = cblk * kCB + c4 * 4;
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
            for (int u = 0; u < NPF; ++u) {
                const int hv = (tid >> 2) + 64 * u;
                if (hv >= HALO) continue;
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if ((pf_ok >> u) & 1u) {   // zero padding applies to the ACTIVATED tensor: outside stays 0
                    v = pre[u];
                    if (a.x_ss) {
                        v.x = __builtin_fmaf(v.x, ss[0], ss[1]); v.y = __builtin_fmaf(v.y, ss[2], ss[3]);
                        v.z = __builtin_fmaf(v.z, ss[4], ss[5]); v.w = __builtin_fmaf(v.w, ss[6], ss[7]);
                    }
                    if (a.x_relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
                    if constexpr (RES) {
                        f32x4 r = prer[u];
                        if (a.res_ss) {
                            r.x = __builtin_fmaf(r.x, rs[0], rs[1]); r.y = __builtin_fmaf(r.y, rs[2], rs[3]);
                            r.z = __builtin_fmaf(r.z, rs[4], rs[5]); r.w = __builtin_fmaf(r.w, rs[6], rs[7]);
                        }
                        if (a.res_relu) { r.x = fmaxf(r.x, 0.f); r.y = fmaxf(r.y, 0.f); r.z = fmaxf(r.z, 0.f); r.w = fmaxf(r.w, 0.f); }
                        v = v + r;
                    }
                    if (a.mat && ((pf_own >> u) & 1u))
                        *reinterpret_cast<f32x4*>(a.mat + pf_off[u] + cblk * kCB) = v;
                }
                *reinterpret_cast<f32x4*>(lds + lds_slot(hv, c4)) = v;
            }
        }
        __syncthreads();

        // ---- 9 taps x G4 k-groups; B operand streamed from L2 (packed: one 1 KB line per wave load) ----
        const f32x4* wb = reinterpret_cast<const f32x4*>(wpk) + (size_t)cblk * (G4 * NF * 64) + lane;
        // B runs BD steps ahead of its use: vmcnt retires in order, so a B load also waits for the (HBM-latency)
        // prefetch words issued before it; A (LDS) runs one step ahead
        constexpr int BD = NTAP == 1 ? 1 : (COUT <= 64 ? 3 : 1), NB = BD + 1;
        f32x4 Bn[NB][NF], An[2][2];
#pragma unroll
        for (int b = 0; b < BD; ++b) {
            const f32x4* w0 = wb + (size_t)(b / G4) * wstep + (b % G4) * (NF * 64);
#pragma unroll
            for (int f = 0; f < NF; ++f) Bn[b][f] = w0[f * 64];
        }
        An[0][0] = *reinterpret_cast<const f32x4*>(lds + lds_slot(hv0, khalf * 2));
        An[0][1] = *reinterpret_cast<const f32x4*>(lds + lds_slot(hv0 + 2 * HS, khalf * 2));
        // next block's words; in the last block every lane re-reads element 0 instead (one cached line, no branch)
        const unsigned nb = (cblk + 1) * kCB, live = cblk + 1 < ncblk ? ~0u : 0u;
#pragma unroll
        for (int s = 0; s < NSTEP; ++s) {
            const int cur = s & 1, nxt = cur ^ 1;
            if (s + 1 < NSTEP) {  // A operands of step s+1
                const int tap = (s + 1) / G4, g = (s + 1) % G4;
                const int voff = NTAP == 9 ? ((tap / 3) * HS + (tap % 3)) * DIL : NTAP == 4 ? (tap >> 1) * HS + (tap & 1) : 0;  // tap offset in halo pixels
                int h0 = hv0;
                asm volatile("" : "+v"(h0));  // keep the swizzled addresses out of long-lived registers
                An[nxt][0] = *reinterpret_cast<const f32x4*>(lds + lds_slot(h0 + voff, khalf * 2 + g));
                An[nxt][1] = *reinterpret_cast<const f32x4*>(lds + lds_slot(h0 + 2 * HS + voff, khalf * 2 + g));
            }
            if (s + BD < NSTEP) {  // B operands of step s+BD
                const int tap = (s + BD) / G4, g = (s + BD) % G4;
                const f32x4* wn = wb + (size_t)tap * wstep + g * (NF * 64);
#pragma unroll
                for (int f = 0; f < NF; ++f) Bn[(s + BD) % NB][f] = wn[f * 64];
            }
            // one word of the NEXT channel block per step: x in steps 2 .. 2+NPF-1, the residual operand after it
            if constexpr (PF_SPREAD) {
                if constexpr (!RES) {   // registers to spare: a scalar branch skips the loads in the last block
                    if (s >= 2 && s < 2 + NPF && live) pre[s - 2] = *reinterpret_cast<const f32x4*>(a.x + pf_off[s - 2] + nb);
                } else {
                    if (s >= 2 && s < 2 + NPF) pre[s - 2] = *reinterpret_cast<const f32x4*>(a.x + ((pf_off[s - 2] + nb) & live));
                }
                if constexpr (RES) {
                    if (s >= 2 + NPF && s < 2 + 2 * NPF)
                        prer[s - 2 - NPF] = *reinterpret_cast<const f32x4*>(a.res + ((pf_off[s - 2 - NPF] + nb) & live));
                }
            } else {   // short step loops (1x1): everything at once
                if (s == 0) {
#pragma unroll
                    for (int u = 0; u < NPF; ++u) pre[u] = *reinterpret_cast<const f32x4*>(a.x + ((pf_off[u] + nb) & live));
                }
                if constexpr (RES) {
                    if (s == 1) {
#pragma unroll
                        for (int u = 0; u < NPF; ++u) prer[u] = *reinterpret_cast<const f32x4*>(a.res + ((pf_off[u] + nb) & live));
                    }
                }
            }
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int m = 0; m < 2; ++m)
#pragma unroll
                    for (int f = 0; f < NF; ++f)
                        acc[m][f] = __builtin_amdgcn_mfma_f32_32x32x2f32(An[cur][m][e], Bn[s % NB][f][e], acc[m][f], 0, 0, 0);
        }
        __syncthreads();
    }

    // ---- epilogue: output (channels-last) + per-channel partial statistics ----
    float s1[NF], s2[NF], bs[NF];
#pragma unroll
    for (int f = 0; f < NF; ++f) { s1[f] = 0.f; s2[f] = 0.f; bs[f] = a.bias ? a.bias[f * 32 + i] : 0.f; }
    if constexpr (EPI == 1) {
        // log_softmax over the COUT columns of each pixel: a pixel (MFMA row) lives in register r of the 32 lanes of its
        // half-wave, one column per lane and fragment -> max / sum over the fragments in the lane, then a 5-step butterfly
        // across the 32 lanes.  Stored planar [N][COUT][H][W]: registers 4q..4q+3 of a lane are 4 consecutive x of one row,
        // i.e. one 16-byte store into the lane's own channel plane.
        const size_t plane = (size_t)a.H * a.W;
#pragma unroll
        for (int m = 0; m < 2; ++m) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float z[4][NF];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float mx = -INFINITY;
#pragma unroll
                    for (int f = 0; f < NF; ++f) { z[e][f] = acc[m][f][4 * q + e] + bs[f]; mx = fmaxf(mx, z[e][f]); }
#pragma unroll
                    for (int o = 1; o < 32; o <<= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
                    float sm = 0.f;
#pragma unroll
                    for (int f = 0; f < NF; ++f) { z[e][f] = z[e][f] - mx; sm += expf(z[e][f]); }
#pragma unroll
                    for (int o = 1; o < 32; o <<= 1) sm += __shfl_xor(sm, o, 64);
                    const float ls = logf(sm);
#pragma unroll
                    for (int f = 0; f < NF; ++f) z[e][f] = z[e][f] - ls;
                }
                const int row = 8 * q + 4 * khalf;               // first of the 4 MFMA rows held in registers 4q..4q+3
                int ry, rx;
                row_to_yx(row, ry, rx);
                const int gy = y0 + wy + 2 * m + ry, gx = x0 + rx;
                if (gy < a.H) {
#pragma unroll
                    for (int f = 0; f < NF; ++f) {
                        float* o = a.planar + ((size_t)n * COUT + f * 32 + i) * plane + (size_t)gy * a.W + gx;
                        if (gx + 3 < a.W && (a.W & 3) == 0) {
                            *reinterpret_cast<f32x4*>(o) = f32x4{z[0][f], z[1][f], z[2][f], z[3][f]};
                        } else {
#pragma unroll
                            for (int e = 0; e < 4; ++e)
                                if (gx + e < a.W) o[e] = z[e][f];
                        }
                    }
                }
            }
        }
        return;
    }
#pragma unroll
    for (int m = 0; m < 2; ++m) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * khalf;  // MFMA C/D row held in register r
            int ry, rx;
            row_to_yx(row, ry, rx);
            const int gy = y0 + wy + 2 * m + ry, gx = x0 + rx;
            if (gy < a.H && gx < a.W) {
                const size_t pix = a.up ? ((size_t)n * (2 * a.H) + 2 * gy + pa) * (2 * a.W) + 2 * gx + pb
                                        : ((size_t)n * a.H + gy) * a.W + gx;
#pragma unroll
                for (int f = 0; f < NF; ++f) {
                    float z = acc[m][f][r] + bs[f];
                    if (a.out_lrelu) z = z > 0.f ? z : 0.01f * z;
                    if (f * 32 + i < a.cout_valid) a.y[pix * a.ldy + a.ycoff + f * 32 + i] = z;
                    s1[f] += z;
                    s2[f] = __builtin_fmaf(z, z, s2[f]);
                }
            }
        }
    }
    if (a.stats) {
        float* red = lds;  // reuse: [4 waves][2*COUT]
#pragma unroll
        for (int f = 0; f < NF; ++f) {
            s1[f] += __shfl_xor(s1[f], 32, 64);
            s2[f] += __shfl_xor(s2[f], 32, 64);
        }
        if (khalf == 0) {
#pragma unroll
            for (int f = 0; f < NF; ++f) {
                red[wv * (2 * COUT) + f * 32 + i] = s1[f];
                red[wv * (2 * COUT) + COUT + f * 32 + i] = s2[f];
            }
        }
        __syncthreads();
        if (tid < 2 * COUT)
            a.stats[(size_t)tile_id * (2 * COUT) + tid] =
                (red[tid] + red[2 * COUT + tid]) + (red[4 * COUT + tid] + red[6 * COUT + tid]);
    }
}

// weights [Cout][Cin][taps] (torch layout, taps = 9 or 27) -> packed [tap][cblk][g][nfrag][lane = khalf*32 + j][4]
//   value = w[cout = nfrag*32 + j][cin = cblk*16 + khalf*8 + g*4 + e][tap]
__global__ void conv_pack_weights_kernel(const float* __restrict__ w, float* __restrict__ wp, int Cin, int Cout, int taps) {
    const int ncblk = Cin / kCB, G4 = kCB / 8, nf = Cout / 32;
    const long total = (long)taps * Cin * Cout;
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    long t = idx;
    const int e = t & 3; t >>= 2;
    const int ln = t & 63; t >>= 6;
    const int nfrag = t % nf; t /= nf;
    const int g = t % G4; t /= G4;
    const int cblk = t % ncblk; const int tap = (int)(t / ncblk);
    const int cout = nfrag * 32 + (ln & 31);
    const int cin = cblk * kCB + (ln >> 5) * (kCB / 2) + g * 4 + e;
    wp[idx] = w[((size_t)cout * Cin + cin) * taps + tap];
}

// Per-channel (sum, sum of squares) partials of a channels-last tensor x [P][C] produced by a vendor-library
// convolution (the stride-2 and 1x1 layers of the trunk): same [workgroup][2C] format as the conv epilogue, so
// bn_finalize serves both.  Thread = one 16-B channel word of a strided pixel slice; LDS tree over the slices.
__global__ __launch_bounds__(256) void nhwc_stats_kernel(const float* __restrict__ x, long P, int C, int pix_per_wg,
                                                         float* __restrict__ stats) {
    __shared__ f32x4 sh[2][256];
    const int tid = threadIdx.x, cq = C >> 2, q = tid % cq, pl = tid / cq, npl = 256 / cq;
    const long p0 = (long)blockIdx.x * pix_per_wg, p1 = (p0 + pix_per_wg < P) ? p0 + pix_per_wg : P;
    f32x4 s1 = {0.f, 0.f, 0.f, 0.f}, s2 = {0.f, 0.f, 0.f, 0.f};
    for (long p = p0 + pl; p < p1; p += npl) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(x + p * C + q * 4);
        s1 += v;
        s2 += v * v;
    }
    sh[0][tid] = s1; sh[1][tid] = s2;
    __syncthreads();
    for (int o = npl >> 1; o > 0; o >>= 1) {
        if (pl < o) { sh[0][tid] += sh[0][tid + o * cq]; sh[1][tid] += sh[1][tid + o * cq]; }
        __syncthreads();
    }
    if (pl == 0) {
        float* o = stats + (size_t)blockIdx.x * 2 * C + q * 4;
        *reinterpret_cast<f32x4*>(o) = sh[0][tid];
        *reinterpret_cast<f32x4*>(o + C) = sh[1][tid];
    }
}

// y = act(x*s+t) [+ act(res*s'+t')] on channels-last [P][C]: the stand-alone form of the loader's prologue, for
// consumers that are not the conv kernel (pooling, the 320-channel concat, the 1x1 head, vendor stride-2 convs).
// y may have a wider pixel stride than C (ldy) so the result can land inside a concat / texel buffer.
__global__ __launch_bounds__(256) void nhwc_act_kernel(const float* __restrict__ x, const float* __restrict__ x_ss,
                                                       int x_relu, const float* __restrict__ res,
                                                       const float* __restrict__ res_ss, int res_relu,
                                                       float* __restrict__ y, long P, int C, int ldy) {
    const int cq = C >> 2;
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= P * cq) return;
    const long p = idx / cq;
    const int c = (int)(idx - p * cq) * 4;
    f32x4 v = *reinterpret_cast<const f32x4*>(x + p * C + c);
    if (x_ss) {
        const float* ss = x_ss + 2 * c;
        v.x = __builtin_fmaf(v.x, ss[0], ss[1]); v.y = __builtin_fmaf(v.y, ss[2], ss[3]);
        v.z = __builtin_fmaf(v.z, ss[4], ss[5]); v.w = __builtin_fmaf(v.w, ss[6], ss[7]);
    }
    if (x_relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
    if (res) {
        f32x4 r = *reinterpret_cast<const f32x4*>(res + p * C + c);
        if (res_ss) {
            const float* ss = res_ss + 2 * c;
            r.x = __builtin_fmaf(r.x, ss[0], ss[1]); r.y = __builtin_fmaf(r.y, ss[2], ss[3]);
            r.z = __builtin_fmaf(r.z, ss[4], ss[5]); r.w = __builtin_fmaf(r.w, ss[6], ss[7]);
        }
        if (res_relu) { r.x = fmaxf(r.x, 0.f); r.y = fmaxf(r.y, 0.f); r.z = fmaxf(r.z, 0.f); r.w = fmaxf(r.w, 0.f); }
        v = v + r;
    }
    *reinterpret_cast<f32x4*>(y + p * ldy + c) = v;
}

// Reduce per-workgroup partials [nwg][2C] to BatchNorm (scale, shift) [C][2] and update the running statistics
// (train mode: biased variance normalises, unbiased variance feeds running_var).  One workgroup per channel.
__global__ __launch_bounds__(256) void bn_finalize_kernel(const float* __restrict__ stats, int nwg, int C, double count,
                                                          const float* __restrict__ gamma, const float* __restrict__ beta,
                                                          float eps, float momentum, float* __restrict__ running_mean,
                                                          float* __restrict__ running_var, float* __restrict__ ss, unsigned int* __restrict__ collapse_count, long long* __restrict__ batches_tracked) {
    __shared__ double sh[2][256];
    const int c = blockIdx.x, tid = threadIdx.x;
    double s1 = 0.0, s2 = 0.0;
    for (int g = tid; g < nwg; g += 256) {
        s1 += (double)stats[(size_t)g * 2 * C + c];
        s2 += (double)stats[(size_t)g * 2 * C + C + c];
    }
    sh[0][tid] = s1; sh[1][tid] = s2;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (tid < o) { sh[0][tid] += sh[0][tid + o]; sh[1][tid] += sh[1][tid + o]; }
        __syncthreads();
    }
    if (tid == 0) bn_finalize_channel(sh[0][0], sh[1][0], count, gamma[c], beta[c], eps, momentum, running_mean, running_var, ss, c, collapse_count);
    if (tid == 0 && c == 0 && batches_tracked) *batches_tracked += 1;      // nn.BatchNorm's num_batches_tracked side effect (one launch less per layer)
}

template <int COUT, int DIL>
static void launch_conv2d(const Conv2dArgs& a, int nwg, hipStream_t st) {
    constexpr int HS = kT2 + 2 * DIL;
    const size_t lds = (size_t)HS * HS * kSV * sizeof(float);
    if (a.res) hipLaunchKernelGGL((conv2d_mfma_kernel<COUT, DIL, true>), dim3(nwg), dim3(256), lds, st, a);
    else       hipLaunchKernelGGL((conv2d_mfma_kernel<COUT, DIL, false>), dim3(nwg), dim3(256), lds, st, a);
}

// R-Net forms: no residual operand, dilation 1; NTAP = 4 (transposed-conv phase) or EPI = 1 (planar log-softmax)
template <int COUT, int NTAP, int EPI>
static void launch_conv2d_ex(const Conv2dArgs& a, int nwg, hipStream_t st, int ny = 1) {
    constexpr int HS = kT2 + 2;
    const size_t lds = (size_t)HS * HS * kSV * sizeof(float);
    hipLaunchKernelGGL((conv2d_mfma_kernel<COUT, 1, false, NTAP, EPI>), dim3(nwg, ny), dim3(256), lds, st, a);
}

// Small-tap forms with the trunk's prologue / statistics epilogue: NTAP = 1 (1x1 convolution) or 4 (2x2 window ending at the
// pixel: a stride-2 3x3 convolution on the space-to-depth image of its input)
template <int COUT, int NTAP>
static void launch_conv2d_taps(const Conv2dArgs& a, int nwg, hipStream_t st) {
    constexpr int HS = kT2 + 2;
    const size_t lds = (size_t)HS * HS * kSV * sizeof(float);
    hipLaunchKernelGGL((conv2d_mfma_kernel<COUT, 1, false, NTAP, 0>), dim3(nwg), dim3(256), lds, st, a);
}

// y[n][y][x][(py*2+px)*C + c] = x[n][2y+py][2x+px][c] (channels >= 4C zero): with it a stride-2 3x3 convolution is a 2x2-tap
// stride-1 convolution (taps dy, dx in {-1, 0}) whose weights are the 3x3 weights re-indexed (row 2y+ky-1: ky = 0 -> dy = -1,
// py = 1; ky = 1 -> dy = 0, py = 0; ky = 2 -> dy = 0, py = 1), the other (tap, phase) pairs zero.
__global__ __launch_bounds__(256) void space_to_depth2_kernel(const float* __restrict__ x, int nchw, float* __restrict__ y,
                                                              int N, int C, int H, int W, int Cp) {
    const int Ho = H >> 1, Wo = W >> 1;
    const long total = (long)N * Ho * Wo * Cp;
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int cp = (int)(idx % Cp);
    long t = idx / Cp;
    const int xo = (int)(t % Wo); t /= Wo;
    const int yo = (int)(t % Ho);
    const int n = (int)(t / Ho);
    float v = 0.f;
    if (cp < 4 * C) {
        const int ph = cp / C, c = cp - ph * C;
        const int yi = 2 * yo + (ph >> 1), xi = 2 * xo + (ph & 1);
        v = nchw ? x[(((size_t)n * C + c) * H + yi) * W + xi] : x[(((size_t)n * H + yi) * W + xi) * C + c];
    }
    y[idx] = v;
}

// The two forms the path uses, 16 bytes per access (round 3; the element-wise kernel above stays for any other shape):
//   channels-last input with C % 4 == 0 (the 32-channel half-resolution map of layer2's first block): one lane = one 16-byte word
//   of the output pixel, read from one phase pixel of the input
__global__ __launch_bounds__(256) void space_to_depth2_cl4_kernel(const float* __restrict__ x, float* __restrict__ y, int N, int C, int H,
                                                                  int W, int Cp) {
    const int Ho = H >> 1, Wo = W >> 1, q4 = Cp >> 2;
    const long total = (long)N * Ho * Wo * q4;
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int cp = (int)(idx % q4) * 4;
    long t = idx / q4;
    const int xo = (int)(t % Wo); t /= Wo;
    const int yo = (int)(t % Ho);
    const int n = (int)(t / Ho);
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (cp < 4 * C) {
        const int ph = cp / C, c = cp - ph * C;
        v = *reinterpret_cast<const float4*>(x + (((size_t)n * H + 2 * yo + (ph >> 1)) * W + 2 * xo + (ph & 1)) * C + c);
    }
    *reinterpret_cast<float4*>(y + idx * 4) = v;
}

//   planar RGB image (C = 3, Cp = 16: the stem's first convolution): one lane = one output pixel = six 8-byte loads (a plane row's
//   two phase columns are adjacent) and four 16-byte stores
__global__ __launch_bounds__(256) void space_to_depth2_rgb_kernel(const float* __restrict__ x, float* __restrict__ y, int N, int H, int W) {
    const int Ho = H >> 1, Wo = W >> 1;
    const long total = (long)N * Ho * Wo;
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int xo = (int)(idx % Wo);
    long t = idx / Wo;
    const int yo = (int)(t % Ho);
    const int n = (int)(t / Ho);
    float2 v[3][2];   // [c][py] = (px 0, px 1)
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int py = 0; py < 2; ++py)
            v[c][py] = *reinterpret_cast<const float2*>(x + (((size_t)n * 3 + c) * H + 2 * yo + py) * W + 2 * xo);
    // output channel (py*2 + px)*3 + c
    float o[16];
#pragma unroll
    for (int py = 0; py < 2; ++py)
#pragma unroll
        for (int c = 0; c < 3; ++c) { o[(py * 2 + 0) * 3 + c] = v[c][py].x; o[(py * 2 + 1) * 3 + c] = v[c][py].y; }
    o[12] = o[13] = o[14] = o[15] = 0.f;
    float4* dst = reinterpret_cast<float4*>(y + idx * 16);
#pragma unroll
    for (int q = 0; q < 4; ++q) dst[q] = make_float4(o[4 * q], o[4 * q + 1], o[4 * q + 2], o[4 * q + 3]);
}

}  // namespace nrgbd

extern "C" int nrgbd_space_to_depth2(const float* x, int nchw, float* y, int N, int C, int H, int W, int Cp, void* stream) {
    using namespace nrgbd;
    if (!x || !y) return NRGBD_E_NULL;
    if (N <= 0 || C <= 0 || H <= 0 || W <= 0 || (H & 1) || (W & 1) || Cp < 4 * C) return NRGBD_E_SHAPE;
    const long total = (long)N * (H / 2) * (W / 2) * Cp;
    const bool aligned = (((uintptr_t)x | (uintptr_t)y) & 15) == 0;
    if (!nchw && (C & 3) == 0 && (Cp & 3) == 0 && aligned)
        hipLaunchKernelGGL(space_to_depth2_cl4_kernel, dim3((unsigned)((total / 4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, y, N, C,
                           H, W, Cp);
    else if (nchw && C == 3 && Cp == 16 && aligned)        // W even: a plane row's phase pair is 8-byte aligned
        hipLaunchKernelGGL(space_to_depth2_rgb_kernel, dim3((unsigned)((total / 16 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, y, N,
                           H, W);
    else
        hipLaunchKernelGGL(space_to_depth2_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, nchw,
                           y, N, C, H, W, Cp);
    NRGBD_CHECK_LAUNCH();
    return NRGBD_OK;
}

extern "C" int nrgbd_conv2d_taps_f32(const float* x, const float* x_ss, int x_relu, const float* w_packed, float* y,
                                     float* stats, int N, int H, int W, int Cin, int Cout, int taps, int in_stride, void* stream) {
    using namespace nrgbd;
    if (!x || !w_packed || !y) return NRGBD_E_NULL;
    if (N <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cin % kCB) return NRGBD_E_SHAPE;
    if (taps != 1 && taps != 4) return NRGBD_E_ARG;
    if (in_stride < 1 || in_stride > 8 || (in_stride != 1 && taps != 1)) return NRGBD_E_ARG;
    if ((long)N * H * W * Cin * in_stride * in_stride >= (1L << 32)) return NRGBD_E_SHAPE;
    Conv2dArgs a{x, x_ss, nullptr, nullptr, nullptr, w_packed, nullptr, y, stats, x_relu, 0, 0, N, H, W, Cin,
                 0, Cout, 0, Cout, 0, 0, 0, nullptr, in_stride};
    const int nwg = ceil_div(W, kT2) * ceil_div(H, kT2) * N;
    hipStream_t st = (hipStream_t)stream;
    if (taps == 1) {
        if (Cout == 32) launch_conv2d_taps<32, 1>(a, nwg, st);
        else if (Cout == 64) launch_conv2d_taps<64, 1>(a, nwg, st);
        else if (Cout == 128) launch_conv2d_taps<128, 1>(a, nwg, st);
        else return NRGBD_E_SHAPE;
    } else {
        if (Cout == 32) launch_conv2d_taps<32, 4>(a, nwg, st);
        else if (Cout == 64) launch_conv2d_taps<64, 4>(a, nwg, st);
        else return NRGBD_E_SHAPE;
    }
    NRGBD_CHECK_LAUNCH();
    return NRGBD_OK;
}

extern "C" int nrgbd_conv2d_workgroups(int N, int H, int W) {
    using namespace nrgbd;
    if (N <= 0 || H <= 0 || W <= 0) return NRGBD_E_SHAPE;
    return ceil_div(W, kT2) * ceil_div(H, kT2) * N;
}

extern "C" int nrgbd_conv_pack_weights(const float* w, float* wp, int Cin, int Cout, int taps, void* stream) {
    using namespace nrgbd;
    if (!w || !wp) return NRGBD_E_NULL;
    if (Cin <= 0 || Cin % kCB || Cout <= 0 || Cout % 32 || (taps != 9 && taps != 27 && taps != 4 && taps != 1)) return NRGBD_E_SHAPE;
    const long total = (long)taps * Cin * Cout;
    hipLaunchKernelGGL(conv_pack_weights_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       w, wp, Cin, Cout, taps);
    NRGBD_CHECK_LAUNCH();
    return NRGBD_OK;
}

extern "C" int nrgbd_conv2d_3x3_f32(const float* x, const float* x_ss, int x_relu, const float* res, const float* res_ss,
                                    int res_relu, float* materialized, const float* w_packed, const float* bias,
                                    int out_lrelu, float* y, float* stats, int N, int H, int W, int Cin, int Cout,
                                    int dilation, void* stream) {
    using namespace nrgbd;
    if (!x || !w_packed || !y) return NRGBD_E_NULL;
    if (N <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cin % kCB) return NRGBD_E_SHAPE;
    if ((long)N * H * W * Cin >= (1L << 32)) return NRGBD_E_SHAPE;  // 32-bit element offsets in the loader
    Conv2dArgs a{x, x_ss, res, res_ss, materialized, w_packed, bias, y, stats, x_relu, res_relu, out_lrelu, N, H, W, Cin,
                 dev_env_int("NRGBD_XCD"), Cout, 0, Cout, 0, 0, 0, nullptr};
    const int nwg = ceil_div(W, kT2) * ceil_div(H, kT2) * N;
    hipStream_t st = (hipStream_t)stream;
    if (dilation == 1 && Cout == 32) launch_conv2d<32, 1>(a, nwg, st);
    else if (dilation == 1 && Cout == 64) launch_conv2d<64, 1>(a, nwg, st);
    else if (dilation == 1 && Cout == 96) launch_conv2d<96, 1>(a, nwg, st);
    else if (dilation == 1 && Cout == 128) launch_conv2d<128, 1>(a, nwg, st);
    else if (dilation == 2 && Cout == 128) launch_conv2d<128, 2>(a, nwg, st);
    else return NRGBD_E_SHAPE;
    NRGBD_CHECK_LAUNCH();
    return NRGBD_OK;
}

extern "C" int nrgbd_conv2d_rnet_f32(const float* x, const float* w_packed, const float* bias, int out_lrelu, float* y,
                                     int ldy, int ycoff, int cout_valid, int mode, int pa, int pb, int N, int H, int W,
                                     int Cin, int Cout, void* stream) {
    using namespace nrgbd;
    if (!x || !w_packed || !y) return NRGBD_E_NULL;
    if (N <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cin % kCB) return NRGBD_E_SHAPE;
    if ((long)N * H * W * Cin >= (1L << 32)) return NRGBD_E_SHAPE;
    if (mode < 0 || mode > 3 || (pa & ~1) || (pb & ~1)) return NRGBD_E_ARG;
    if (mode != 2 && (cout_valid <= 0 || cout_valid > Cout || ycoff < 0 || ldy < ycoff + cout_valid)) return NRGBD_E_SHAPE;
    Conv2dArgs a{x, nullptr, nullptr, nullptr, nullptr, w_packed, bias, y, nullptr, 0, 0, out_lrelu, N, H, W, Cin,
                 0, ldy, ycoff, cout_valid, mode == 1 ? 1 : (mode == 3 ? 2 : 0), pa, pb, mode == 2 ? y : nullptr};
    const int nwg = ceil_div(W, kT2) * ceil_div(H, kT2) * N;
    hipStream_t st = (hipStream_t)stream;
    if (mode == 0) {          // 3x3 convolution (+ bias, LeakyReLU), output anywhere inside a wider pixel
        if (Cout == 64) launch_conv2d_ex<64, 9, 0>(a, nwg, st);
        else if (Cout == 96) launch_conv2d_ex<96, 9, 0>(a, nwg, st);
        else if (Cout == 128) launch_conv2d_ex<128, 9, 0>(a, nwg, st);
        else return NRGBD_E_SHAPE;
    } else if (mode == 1) {   // one phase of ConvTranspose2d(k4, s2, p1)
        if (Cout == 64) launch_conv2d_ex<64, 4, 0>(a, nwg, st);
        else return NRGBD_E_SHAPE;
    } else if (mode == 3) {   // all four phases, w_packed = [phase (pa, pb) = (0,0), (0,1), (1,0), (1,1)][4 taps * Cin * Cout]
        if (Cout == 64) launch_conv2d_ex<64, 4, 0>(a, nwg, st, 4);
        else return NRGBD_E_SHAPE;
    } else {                  // last layer: bias + log_softmax over the channels, planar output
        if (Cout == 64) launch_conv2d_ex<64, 9, 1>(a, nwg, st);
        else if (Cout == 128) launch_conv2d_ex<128, 9, 1>(a, nwg, st);
        else return NRGBD_E_SHAPE;
    }
    NRGBD_CHECK_LAUNCH();
    return NRGBD_OK;
}

// R-Net input assembly (Refine.py:88): x0[p][0..D-1] = exp(dpv_log[d][p]) (the caller's torch.exp), x0[p][D..D+Cf-1] =
// quarter-resolution features (channels-last [P][Cf] or planar [Cf][P]); one pass instead of exp + permute + cat.
namespace nrgbd {
__global__ __launch_bounds__(256) void rnet_pack_kernel(const float* __restrict__ dpv_log, const float* __restrict__ feat,
                                                        int feat_planar, float* __restrict__ out, int D, int Cf, long P) {
    __shared__ float tile[64][65];
    const long p0 = (long)blockIdx.x * 64;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int C = D + Cf;
    for (int c0 = 0; c0 < C; c0 += 64) {
        // a 64-channel x 64-pixel block through LDS: planar reads coalesced along p, channels-last writes along c
        for (int c = wv; c < 64 && c0 + c < C; c += 4) {
            const int ch = c0 + c;
            const long p = p0 + lane;
            float v = 0.f;
            if (p < P) {
                if (ch < D) v = expf(dpv_log[(size_t)ch * P + p]);
                else v = feat_planar ? feat[(size_t)(ch - D) * P + p] : feat[(size_t)p * Cf + (ch - D)];
            }
            tile[c][lane] = v;
        }
        __syncthreads();
        for (int t = tid; t < 64 * 64; t += 256) {
            const int px = t >> 6, c = t & 63;
            if (p0 + px < P && c0 + c < C) out[(size_t)(p0 + px) * C + c0 + c] = tile[c][px];
        }
        __syncthreads();
    }
}
}  // namespace nrgbd

extern "C" int nrgbd_rnet_pack(const float* dpv_log, const float* feat, int feat_planar, float* out, int D, int Cf, long P,
                               void* stream) {
    if (!dpv_log || !feat || !out) return NRGBD_E_NULL;
    if (D <= 0 || Cf <= 0 || P <= 0) return NRGBD_E_SHAPE;
    hipLaunchKernelGGL(nrgbd::rnet_pack_kernel, dim3((unsigned)((P + 63) / 64)), dim3(256), 0, (hipStream_t)stream, dpv_log,
                       feat, feat_planar, out, D, Cf, P);
    NRGBD_CHECK_LAUNCH();
    return NRGBD_OK;
}

extern "C" int nrgbd_nhwc_stats_workgroups(long P) {
    if (P <= 0) return NRGBD_E_SHAPE;
    const long per = 2048;  // pixels per workgroup
    return (int)((P + per - 1) / per);
}

extern "C" int nrgbd_nhwc_stats(const float* x, long P, int C, float* stats, void* stream) {
    using namespace nrgbd;
    if (!x || !stats) return NRGBD_E_NULL;
    if (P <= 0 || C < 4 || (C & 3) || 256 % (C >> 2) || ((256 / (C >> 2)) & (256 / (C >> 2) - 1))) return NRGBD_E_SHAPE;
    const int nwg = nrgbd_nhwc_stats_workgroups(P);
    hipLaunchKernelGGL(nhwc_stats_kernel, dim3(nwg), dim3(256), 0, (hipStream_t)stream, x, P, C, 2048, stats);
    NRGBD_CHECK_LAUNCH();
    return NRGBD_OK;
}

extern "C" int nrgbd_nhwc_act(const float* x, const float* x_ss, int x_relu, const float* res, const float* res_ss,
                              int res_relu, float* y, long P, int C, int ldy, void* stream) {
    using namespace nrgbd;
    if (!x || !y) return NRGBD_E_NULL;
    if (P <= 0 || C < 4 || (C & 3) || ldy < C || (ldy & 3)) return NRGBD_E_SHAPE;
    const long total = P * (C >> 2);
    hipLaunchKernelGGL(nhwc_act_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, x_ss,
                       x_relu, res, res_ss, res_relu, y, P, C, ldy);
    NRGBD_CHECK_LAUNCH();
    return NRGBD_OK;
}

extern "C" int nrgbd_bn_finalize(const float* stats, int num_workgroups, int C, long count, const float* gamma,
                                 const float* beta, float eps, float momentum, float* running_mean, float* running_var,
                                 float* scale_shift, unsigned int* collapse_count, long long* batches_tracked, void* stream) {
    using namespace nrgbd;
    if (!stats || !gamma || !beta || !scale_shift) return NRGBD_E_NULL;
    if (num_workgroups <= 0 || count <= 0 || C <= 0) return NRGBD_E_SHAPE;
    if ((running_mean == nullptr) != (running_var == nullptr)) return NRGBD_E_NULL;
    hipLaunchKernelGGL(bn_finalize_kernel, dim3(C), dim3(256), 0, (hipStream_t)stream, stats, num_workgroups, C,
                       (double)count, gamma, beta, eps, momentum, running_mean, running_var, scale_shift, collapse_count, batches_tracked);
    NRGBD_CHECK_LAUNCH();
    return NRGBD_OK;
}
