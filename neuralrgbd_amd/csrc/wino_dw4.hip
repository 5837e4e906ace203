// wino_dw4.hip — the K-Net's 64 -> 64 3x3x3 convolutions (models/basic.py:71-94) with F(4, 3) ALONG THE DEPTH AXIS on top of the
// in-plane F(2x2, 3x3): 6 transform points per FOUR output slices = 6 multiplies per output voxel (wino_dw.hip's F(2, 3): 8; direct: 27).
//
// Why (profiles/r6_wino_d4_probe.txt).  wino_dw.hip runs at 0.67 of the fp32 matrix peak with the matrix pipe and the
// producers' issue slots as co-limits; the lever left is fewer multiplies.  F(4x4, 3x3) in the plane was rejected on numerics in round 5
// (3.9x further from float64 than the direct convolution).  F(4, 3) along depth ONLY applies the ill-conditioned transform once, and with
// the interpolation points (0, +-1/2, +-3/2, inf) the whole K-Net sits 1.24x as far from float64 as the direct convolution (today's form:
// 0.92x; acceptance 1.25x; oracle/wino_d4_eval.py).
//
//   input    d_j = act(x[z0 - 1 + j]),  j = 0..5                       (z0 = first of the four output slices of a tile)
//   depth    D_t = sum_j Bd[t][j] d_j,  Bd = [ 9/16 0 -5/2 0 1 0 | 0 -9/8 -9/4 1/2 1 0 | 0 9/8 -9/4 -1/2 1 0 | 0 -3/8 -1/4 3/2 1 0 |
//                                               0 3/8 -1/4 -3/2 1 0 | 0 9/16 0 -5/2 0 1 ]            (3 or 4 slices per D_t)
//   plane    V_t = B^T D_t B per 4x4 patch                                                    (as wino_pc.hip / wino_dw.hip)
//   weights  U_t = sum_kd Gd[t][kd] (G g_kd G^T),  Gd = [16/9 0 0 | -1 -1/2 -1/4 | -1 1/2 -1/4 | 1/9 1/6 1/4 | 1/9 -1/6 1/4 | 0 0 1]
//   product  M_t = sum_ci V_t U_t                                                             (the MFMAs: 6 x Cin/16 stages per tile)
//   output   y[z0]   = M0 + (M1 + M2) + (M3 + M4)          y[z0+1] = 1/2 (M1 - M2) + 3/2 (M3 - M4)
//            y[z0+2] = 1/4 (M1 + M2) + 9/4 (M3 + M4)       y[z0+3] = 1/8 (M1 - M2) + 27/8 (M3 - M4) + M5
//
// Same persistent producer / consumer organisation, tile geometry, LDS images, shared strips and early stage barrier as wino_dw.hip (read
// its header first).  What differs:
//   * a tile is 8x16 pixels x FOUR depth slices; phases run in the order t = 1, 2, 3, 4, 0, 5 so that the fold needs few live values:
//       after M1: A = M1 | after M2: A = S12 = M1 + M2, B = D12 = M1 - M2 | after M3: C = M3 | after M4: S34 = C + M4, D34 = C - M4,
//       slices z0+1 and z0+2 are COMPLETE and stored, A = S12 + S34, B = D12 / 8 + 27/8 D34 | after M0: slice z0 = A + M0 |
//       after M5: slice z0+3 = B + M5.
//     A and B are the two 32 KB LDS stashes wino_dw.hip has; C — live for one phase only — does not fit the LDS (it would be the third of
//     three: 185 KB) and goes through a per-workgroup 32 KB scratch in global memory that the wave itself wrote (8 stores + 8 loads of 1 KB
//     per tile: 0.5 % of the tile's issue slots; the 8 MB of all workgroups stay in the L2s).
//   * producers: a stage combines 3 or 4 slices with rational coefficients.  The unit of prefetch stays one (slice, channel block) in one of
//     four register sets (the fourth empty for t = 0 and t = 5), all requested one stage ahead; the stage's combination sum_k c_k act(x_k)
//     is formed in registers and the strip written once (wino_dw.hip publishes unit A and read-modify-writes the strip for unit B).
//   * only the two forms the K-Net uses behind its materialise passes: IDENT (x as it is) and CLAMP (relu(x * s + t) as a clamped FMA).
// Work per four output slices: 24 stages (wino_dw: 32); producer units 88 (64): 5.5 per output slice instead of 4, MFMAs 0.75x.
#include <type_traits>

#include "wino_pc.hpp"

namespace nrgbd {

constexpr int kD4StashWave = 2 * 8 * 64 * 4;   // floats of one consumer wave's two LDS stashes: [2][8 words][64 lanes][4]
constexpr int kD4MaxCin = 512;
constexpr int kD4NBuf = 2;
constexpr int kD4ShRows = kPcTH + 2;
constexpr int kD4ShStrip = kD4ShRows * kPcRawW * kCB;      // floats of one shared strip: [10 rows][20 pixels][16] = 12.8 KB
constexpr int kD4ShItems = kD4ShRows * 18 * 4;             // 720 (row, column, 16-byte word) items of a unit
constexpr int kD4NPF = 3;                                  // items per producer lane and unit
constexpr int kD4ScratchWave = 8 * 64 * 4;                 // floats of one consumer wave's global scratch (stash C)

// depth-transform index of phase p (execution order) and the unit slots of index t: (input slice j = 0..5 relative to z0 - 1, coefficient)
__device__ __forceinline__ int d4_t(int p) { return p < 4 ? p + 1 : (p == 4 ? 0 : 5); }
__device__ __forceinline__ int d4_nslot(int t) { return (t == 0 || t == 5) ? 3 : 4; }
__device__ __forceinline__ int d4_j(int t, int k) {
    if (t == 0) return k == 0 ? 4 : (k == 1 ? 0 : 2);
    if (t == 5) return k == 0 ? 5 : (k == 1 ? 1 : 3);
    return k == 0 ? 4 : k;                         // t = 1..4: slices 4 (coefficient 1), 1, 2, 3
}
__device__ __forceinline__ float d4_c(int t, int k) {
    if (k == 0) return 1.f;
    if (t == 0 || t == 5) return k == 1 ? 0.5625f : -2.5f;
    const float s = (t & 1) ? -1.f : 1.f;          // t = 1, 3: the odd part enters with a minus sign
    if (t <= 2) return k == 1 ? s * 1.125f : (k == 2 ? -2.25f : -s * 0.5f);
    return k == 1 ? s * 0.375f : (k == 2 ? -0.25f : -s * 1.5f);
}

struct D4Tile { int z0, y0, x0, cg, row0; };   // row0: statistics row of slice z0 (slice z0 + k: row0 + k)

__device__ __forceinline__ D4Tile d4_decode(int t, const WinoPcArgs& a) {
    D4Tile r;
    const int ncg = a.Cout >> 6;
    const int tiles_x = (a.W + kPcTW - 1) / kPcTW;
    const int row = t / ncg;
    r.cg = t - row * ncg;
    t = row;
    const int nquad = a.N >> 2;
    const int zq = t % nquad; t /= nquad;       // depth fastest: list neighbours share two of their six input slices
    const int tx = t % tiles_x, ty = t / tiles_x;
    r.z0 = 4 * zq;
    r.y0 = ty * kPcTH; r.x0 = tx * kPcTW;
    r.row0 = (ty * tiles_x + tx) * a.N + r.z0;
    return r;
}

// channel block of the i-th stage of phase position p: odd positions sweep the blocks backwards (see wino_dw.hip dw_cb)
#ifndef NRGBD_D4_SERP
#define NRGBD_D4_SERP 1   // 0: experimental A/B builds only (build.build_variant)
#endif
__device__ __forceinline__ int d4_cb(int p, int i, int ncb) { return (NRGBD_D4_SERP && (p & 1)) ? ncb - 1 - i : i; }

struct WinoD4Args {
    WinoPcArgs b;       // x, x_ss, wp, y, stats, N, H, W, Cin, Cout, ntiles, rows, x_unit (res / mat / bias unused)
    float* scratch;     // [workgroups][4 consumer waves][8][64][4] floats: stash C
};

template <bool IDENT, bool CLAMP>
__global__ __launch_bounds__(512) void conv_wino_dw4_kernel(const WinoD4Args aa) {
    const WinoPcArgs& a = aa.b;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* Vb = lds;                                   // [2][16 xi][32 tiles][16]
    float* rawb = lds + kD4NBuf * kPcV;                // [2][10 rows][20 pixels][16] shared strips
    float* stashb = rawb + 2 * kD4ShStrip;             // [4 consumer waves][2 stashes][8][64][4]
    float* ssl = stashb + 4 * kD4StashWave;            // [Cin][2] (scale, shift) of x, pre-paired

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wv = wave & 3;
    const int ncb = a.Cin / kCB;
    const int NS = 6 * ncb;                            // stages per tile (four output slices)

    int first, step, end;
    {
        const int G = (int)gridDim.x, b = (int)blockIdx.x;
        if ((G & 7) == 0) {
            const int xc = b & 7;
            first = (int)(((long)a.ntiles * xc) >> 3) + (b >> 3);
            end = (int)(((long)a.ntiles * (xc + 1)) >> 3);
            step = G >> 3;
        } else { first = b; end = a.ntiles; step = G; }
    }
    if (first >= end) return;                  // a workgroup without tiles (uniform)
    const int count = (end - first + step - 1) / step;
    const unsigned plane = (unsigned)((size_t)a.H * a.W * a.Cin);
    for (int i = tid; i < 2 * a.Cin; i += 512) {
        const int j = pc_ss_slot(i);
        ssl[j] = (a.x_ss ? a.x_ss[i] : ((i & 1) ? 0.f : 1.f)) * (CLAMP ? a.x_unit : 1.f);
    }
    __syncthreads();

    if (wave >= 4) {
        // =========================================== consumer: 16 output channels x 16 xi x 32 tiles, one M_t at a time ========
        const int kq = lane >> 4, jj = lane & 15;
        f32x4 acc[16][2];
        const int a0 = pc_slot(0, jj, kq), a1 = pc_slot(0, 16 + jj, kq);
        const f32x4* wbase = reinterpret_cast<const f32x4*>(a.wp) + wv * 64 + lane;
        const unsigned lane_yoff = (unsigned)jj + (unsigned)((2 * (kq >> 1)) * a.W + 8 * (kq & 1)) * (unsigned)a.Cout;
        const size_t wgroup = (size_t)NS * 16 * 256;
        f32x4* stashA = reinterpret_cast<f32x4*>(stashb + wv * kD4StashWave) + lane;     // word i at stashA[i * 64]
        f32x4* stashB = stashA + 8 * 64;
        // stash C: the wave's 8 KB of the global scratch through a buffer descriptor (uniform base in SGPRs + the lane's 32-bit byte offset:
        // no 64-bit per-lane pointer kept alive across the tile loop — the register file has none to spare)
        const __amdgpu_buffer_rsrc_t stashC = pc_rsrc(reinterpret_cast<const char*>(aa.scratch) +
                                                      ((size_t)blockIdx.x * 4 + wv) * kD4ScratchWave * sizeof(float));
        const int laneC = lane * 16;

        D4Tile tl = d4_decode(first, a);
        const f32x4* wt = wbase + (size_t)tl.cg * wgroup;
        f32x4 Bn[kPcNB], An[2][2];
#pragma unroll
        for (int b = 0; b < kPcBD; ++b) Bn[b] = wt[b * 256];
        __syncthreads();                               // the producers publish stage 0 (transformed one iteration later)
        __syncthreads();                               // producers finish stage 0
        int buf = 0;
        An[0][0] = *reinterpret_cast<const f32x4*>(Vb + a0);
        An[0][1] = *reinterpret_cast<const f32x4*>(Vb + a1);
        float neg1 = -1.f;
        asm volatile("" : "+v"(neg1));
        const f32x2 n1 = {neg1, neg1};

        for (int it = 0; it < count; ++it) {
            const int tnext = first + (it + 1 < count ? it + 1 : it) * step;
            const D4Tile tn = d4_decode(tnext, a);
            const f32x4* wt_next = wbase + (size_t)tn.cg * wgroup;
            const int co = tl.cg * 64 + wv * 16 + jj;
            // one phase = the Cin/16 stages of position P (depth-transform index t = 1, 2, 3, 4, 0, 5), then its fold; the six phases are
            // separate straight-line instantiations so that the accumulators stay in fixed registers.  The weight stream is packed in
            // EXECUTION order: stage s = P * ncb + channel block.
            auto phase = [&](auto p_tag) __attribute__((always_inline)) {
                constexpr int P = decltype(p_tag)::value;
                for (int cb = 0; cb < ncb; ++cb) {
                    const int s = P * ncb + d4_cb(P, cb, ncb);
                    const float* Vc = Vb + buf * kPcV;
                    const int nbuf = buf ^ 1;
                    const float* Vn = Vb + nbuf * kPcV;
                    const f32x4* wcur = wt + (size_t)s * (16 * 256);
                    const f32x4* wnx = cb + 1 < ncb ? wt + (size_t)(P * ncb + d4_cb(P, cb + 1, ncb)) * (16 * 256)
                                       : (P < 5 ? wt + (size_t)((P + 1) * ncb + d4_cb(P + 1, 0, ncb)) * (16 * 256) : wt_next);
                    auto body = [&](auto first_tag) __attribute__((always_inline)) {
                        constexpr bool FIRST = decltype(first_tag)::value;
                        const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                        for (int xi = 0; xi < 16; ++xi) {
                            const int cur = xi & 1, nxt = cur ^ 1;
                            if (xi + 1 < 16) {
                                An[nxt][0] = *reinterpret_cast<const f32x4*>(Vc + a0 + (xi + 1) * (kPcTiles * kCB));
                                An[nxt][1] = *reinterpret_cast<const f32x4*>(Vc + a1 + (xi + 1) * (kPcTiles * kCB));
                            }
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                acc[xi][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(An[cur][0][e], Bn[xi % kPcNB][e],
                                                                                  FIRST && e == 0 ? zero4 : acc[xi][0], 0, 0, 0);
                                acc[xi][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(An[cur][1][e], Bn[xi % kPcNB][e],
                                                                                  FIRST && e == 0 ? zero4 : acc[xi][1], 0, 0, 0);
                                if (e == NRGBD_WPOS) Bn[(xi + kPcBD) % kPcNB] = xi + kPcBD < 16 ? wcur[(xi + kPcBD) * 256] : wnx[(xi + kPcBD - 16) * 256];
                                __builtin_amdgcn_sched_barrier(0);
                            }
                            if (xi == 14) {            // EARLY stage barrier (wino_dw.hip)
                                __syncthreads();
                                An[0][0] = *reinterpret_cast<const f32x4*>(Vn + a0);
                                An[0][1] = *reinterpret_cast<const f32x4*>(Vn + a1);
                                __builtin_amdgcn_sched_barrier(0);
                            }
                        }
                    };
                    if (cb == 0) body(std::true_type{}); else body(std::false_type{});
                    buf = nbuf;
                }
                // ---- end of phase: plane inverse transform of M_t (A^T . A: 32 values per lane) and the depth fold (file header).
                // lane (kq, jj): output channel co = 16 wv + jj; register r of row block m = Winograd tile 16 m + 4 kq + r; word (m, rp, aa) =
                // output row 2 (tile row) + aa, tiles r = 2rp (.x of a pair) and 2rp + 1 (.y), columns 2 (tile column) + {0: o0, 1: o1}
                {
                    constexpr int NEMIT = P == 3 ? 2 : (P >= 4 ? 1 : 0);           // slices completed by this phase
                    constexpr int ZS0 = P == 3 ? 1 : (P == 4 ? 0 : 3);              // the (first) one
                    f32x2 S1[2] = {{0.f, 0.f}, {0.f, 0.f}}, S2[2] = {{0.f, 0.f}, {0.f, 0.f}};
                    float* ybase = a.y + (((size_t)tl.z0 * a.H + tl.y0) * a.W + tl.x0) * a.Cout + tl.cg * 64 + wv * 16;
                    const size_t zstride = (size_t)a.H * a.W * a.Cout;
#pragma unroll
                    for (int m = 0; m < 2; ++m) {
#pragma unroll
                        for (int rp = 0; rp < 2; ++rp) {
                            f32x2 tr[2][4];
#pragma unroll
                            for (int xx = 0; xx < 4; ++xx) {
                                const f32x2 m0 = rp ? acc[0 + xx][m].hi : acc[0 + xx][m].lo, m1 = rp ? acc[4 + xx][m].hi : acc[4 + xx][m].lo;
                                const f32x2 m2 = rp ? acc[8 + xx][m].hi : acc[8 + xx][m].lo, m3 = rp ? acc[12 + xx][m].hi : acc[12 + xx][m].lo;
                                tr[0][xx] = (m0 + m1) + m2;
                                tr[1][xx] = __builtin_elementwise_fma(m3, n1, __builtin_elementwise_fma(m2, n1, m1));   // (m1 - m2) - m3
                            }
#pragma unroll
                            for (int aa2 = 0; aa2 < 2; ++aa2) {
                                const int wi = (m * 2 + rp) * 2 + aa2;
                                const f32x2 o0 = (tr[aa2][0] + tr[aa2][1]) + tr[aa2][2];
                                const f32x2 o1 = __builtin_elementwise_fma(tr[aa2][3], n1, __builtin_elementwise_fma(tr[aa2][2], n1, tr[aa2][1]));
                                const f32x4 o = __builtin_shufflevector(o0, o1, 0, 1, 2, 3);
                                // a completed word of output slice z0 + ZS goes out at once (short live ranges: the register file is full
                                // here — 128 accumulators + weight ring + operands): stores + the slice's partial statistics (slot Q)
                                auto emit = [&](auto zs_tag, auto q_tag, const f32x4 v) __attribute__((always_inline)) {
                                    constexpr int ZS = decltype(zs_tag)::value, Q = decltype(q_tag)::value;
                                    float* oa = ybase + (size_t)ZS * zstride + ((size_t)(4 * m + aa2) * a.W + (size_t)(2 * (2 * rp))) * a.Cout;       // tile r = 2 rp
                                    float* ob = ybase + (size_t)ZS * zstride + ((size_t)(4 * m + aa2) * a.W + (size_t)(2 * (2 * rp + 1))) * a.Cout;   // tile r + 1
                                    oa[lane_yoff] = v.x; oa[lane_yoff + a.Cout] = v.z;
                                    ob[lane_yoff] = v.y; ob[lane_yoff + a.Cout] = v.w;
                                    S1[Q] = (S1[Q] + v.lo) + v.hi;
                                    S2[Q] = __builtin_elementwise_fma(v.hi, v.hi, __builtin_elementwise_fma(v.lo, v.lo, S2[Q]));
                                };
                                using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>;
                                using I2 = std::integral_constant<int, 2>; using I3 = std::integral_constant<int, 3>;
                                if constexpr (P == 0) {                    // M1
                                    stashA[wi * 64] = o;
                                } else if constexpr (P == 1) {             // M2: S12, D12
                                    const f32x4 m1v = stashA[wi * 64];
                                    stashA[wi * 64] = m1v + o;
                                    stashB[wi * 64] = m1v - o;
                                } else if constexpr (P == 2) {             // M3 -> the global scratch (read back one phase later by this lane)
                                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i32x4_t, o), stashC, laneC, wi * 1024, 0);
                                } else if constexpr (P == 3) {             // M4: S34, D34; slices z0+1, z0+2 complete
                                    const f32x4 m3v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(stashC, laneC, wi * 1024, 0));
                                    {
                                        const f32x4 d34 = m3v - o, d12 = stashB[wi * 64];
                                        stashB[wi * 64] = 0.125f * d12 + 3.375f * d34;     // + M5 -> y[z0+3]
                                        emit(I1{}, I0{}, 0.5f * d12 + 1.5f * d34);         // y[z0+1]
                                    }
                                    __builtin_amdgcn_sched_barrier(0);
                                    {
                                        const f32x4 s34 = m3v + o, s12 = stashA[wi * 64];
                                        stashA[wi * 64] = s12 + s34;                       // + M0 -> y[z0]
                                        emit(I2{}, I1{}, 0.25f * s12 + 2.25f * s34);       // y[z0+2]
                                    }
                                    __builtin_amdgcn_sched_barrier(0);
                                } else if constexpr (P == 4) {             // M0
                                    emit(I0{}, I0{}, stashA[wi * 64] + o);
                                } else {                                   // M5
                                    emit(I3{}, I0{}, stashB[wi * 64] + o);
                                }
                            }
                            __builtin_amdgcn_sched_barrier(0);      // one (m, rp) group at a time (register pressure, wino_dw.hip)
                        }
                    }
                    if constexpr (NEMIT >= 1) {
                        if (a.stats) {   // the wave owns its 16 channels: reduce over the 4 lanes (kq) that share a channel
#pragma unroll
                            for (int q = 0; q < NEMIT; ++q) {
                                float s1 = S1[q].x + S1[q].y, s2 = S2[q].x + S2[q].y;
                                s1 += __shfl_xor(s1, 16, 64); s2 += __shfl_xor(s2, 16, 64);
                                s1 += __shfl_xor(s1, 32, 64); s2 += __shfl_xor(s2, 32, 64);
                                if (kq == 0) {
                                    const int row = tl.row0 + ZS0 + q;
                                    a.stats[(size_t)co * a.rows + row] = s1;
                                    a.stats[(size_t)(a.Cout + co) * a.rows + row] = s2;
                                }
                            }
                        }
                    }
                }
            };
            phase(std::integral_constant<int, 0>{});
            phase(std::integral_constant<int, 1>{});
            phase(std::integral_constant<int, 2>{});
            phase(std::integral_constant<int, 3>{});
            phase(std::integral_constant<int, 4>{});
            phase(std::integral_constant<int, 5>{});
            tl = tn;
            wt = wt_next;
        }
    } else {
        // =========================================== producer: tile row pw (8 Winograd tiles) ===========================
        const int pw = wv;
        constexpr int kItems = kD4ShItems;
        float* raw = rawb;                                  // the strip this iteration PUBLISHES into (set per iteration)
        const float* rawT = rawb;                           // ... and the one it TRANSFORMS from
        const int w4 = lane & 3;
        auto item_id = [&](int u) { return 192 * pw + lane + 64 * u; };
        auto item_rr = [&](int u) { return (item_id(u) >> 2) / 18; };
        auto item_cp = [&](int u) { const int pi = item_id(u) >> 2; return pi - (pi / 18) * 18; };
        auto item_col = [&](int u) { const int cp = item_cp(u); return cp < 9 ? 2 * cp : 2 * cp - 17; };
        int wr_off[kD4NPF];
#pragma unroll
        for (int u = 0; u < kD4NPF; ++u) {
            const int item = item_id(u), e = (item - kItems) >> 2;   // lanes without an item write a zero into a pad pixel (columns 18, 19)
            wr_off[u] = item < kItems ? (item_rr(u) * kPcRawW + item_cp(u)) * kCB + w4 * 4
                                      : ((e >> 1) * kPcRawW + 18 + (e & 1)) * kCB + w4 * 4;
        }
        const int tword = lane & 3, txl = ((lane >> 5) << 2) | ((lane >> 2) & 3), thalf = (lane >> 4) & 1;
        const int ttile = pw * 8 + txl;
        const int rdc = txl * kCB + tword * 4 + 2 * pw * kPcRawW * kCB;   // the tile row's halo rows start at strip row 2 pw
        const int rdR0 = (thalf ? 2 : 0) * kPcRawW * kCB + rdc, rdR1 = (thalf ? 1 : 2) * kPcRawW * kCB + rdc,
                  rdR2 = (thalf ? 3 : 1) * kPcRawW * kCB + rdc;
        const float sg = thalf ? -1.f : 1.f;
        float m1 = -1.f;
        asm volatile("" : "+v"(m1));

        unsigned cur_off[kD4NPF], nxt_off[kD4NPF];   // BYTE offsets inside a slice
        float cur_keep[kD4NPF], nxt_keep[kD4NPF];
        auto setup = [&](const D4Tile& tt, unsigned (&b_off)[kD4NPF], float (&b_keep)[kD4NPF]) __attribute__((always_inline)) {
#pragma unroll
            for (int u = 0; u < kD4NPF; ++u) {
                const int hy = item_rr(u), hx = item_col(u);
                const int gy = tt.y0 + hy - 1, gx = tt.x0 + hx - 1;
                const bool in = item_id(u) < kItems && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
                b_off[u] = 4u * (in ? (unsigned)(((size_t)gy * a.W + gx) * a.Cin + w4 * 4) : (unsigned)(w4 * 4));
                b_keep[u] = in ? 1.f : 0.f;
            }
        };
        struct Regs { f32x4 pre[kD4NPF]; };
        D4Tile tl = d4_decode(first, a), tn = tl;
        // raw words of one unit = (slice z0 - 1 + j, channel block cb) -> registers; nx: of the NEXT tile
        auto issue = [&](bool nx, int j, int cb, Regs& r) __attribute__((always_inline)) {
            const int tz = (nx ? tn.z0 : tl.z0) - 1 + j;
#ifdef NRGBD_D4_FAKEZ        // timing experiment only (results invalid): every unit reads slices 0..3 -> the input stays in the L2
            const int z = __builtin_amdgcn_readfirstlane(tz & 3);
#else
            const int z = __builtin_amdgcn_readfirstlane(min(max(tz, 0), a.N - 1));    // clamped: an outside slice is not used when published
#endif
            const size_t base = ((size_t)z * plane + (size_t)(__builtin_amdgcn_readfirstlane(cb) * kCB)) * sizeof(float);
            const __amdgpu_buffer_rsrc_t xb = pc_rsrc(reinterpret_cast<const char*>(a.x) + base);
#pragma unroll
            for (int u = 0; u < kD4NPF; ++u) r.pre[u] = pc_bload(xb, nx ? nxt_off[u] : cur_off[u]);
        };
        setup(tl, cur_off, cur_keep);
        // one register set per unit slot of a stage, refilled with the same slot of the NEXT stage right after it was published: a whole
        // stage for the load to land (two sets alternating inside the stage left one unit of work between request and use)
        Regs set0, set1, set2, set3;
        f32x4 ssw[2] = {{1.f, 1.f, 0.f, 0.f}, {1.f, 1.f, 0.f, 0.f}};     // (scale, shift) pairs of the stage's channel block (pre-paired table)
        {
            const int t0 = d4_t(0), cb0 = d4_cb(0, 0, ncb);
            issue(false, d4_j(t0, 0), cb0, set0);
            issue(false, d4_j(t0, 1), cb0, set1);
            issue(false, d4_j(t0, 2), cb0, set2);
            issue(false, d4_j(t0, 3 < d4_nslot(t0) ? 3 : 0), cb0, set3);
        }
        int qbuf = 0;
        bool has_next = false;

        // normalise / activate one unit's words in place (registers): r.pre[i] <- act(x * s + t)
        auto activate = [&](Regs& r) __attribute__((always_inline)) {
            if constexpr (IDENT) return;
            const f32x2 sc01 = ssw[0].lo, sh01 = ssw[0].hi, sc23 = ssw[1].lo, sh23 = ssw[1].hi;
            f32x2 lo[kD4NPF], hi[kD4NPF];
            if constexpr (CLAMP) {
#pragma unroll
                for (int i = 0; i < kD4NPF; ++i) {
                    lo[i] = pk_fma_clamp01(r.pre[i].lo, sc01, sh01);
                    hi[i] = pk_fma_clamp01(r.pre[i].hi, sc23, sh23);
                }
            } else {
#pragma unroll
                for (int i = 0; i < kD4NPF; ++i) {
                    lo[i] = __builtin_elementwise_fma(r.pre[i].lo, sc01, sh01);
                    hi[i] = __builtin_elementwise_fma(r.pre[i].hi, sc23, sh23);
                }
                __builtin_amdgcn_sched_barrier(0);
                if (a.x_relu) {
#pragma unroll
                    for (int i = 0; i < kD4NPF; ++i) { lo[i].x = relu1(lo[i].x); lo[i].y = relu1(lo[i].y); hi[i].x = relu1(hi[i].x); hi[i].y = relu1(hi[i].y); }
                }
            }
#pragma unroll
            for (int i = 0; i < kD4NPF; ++i) r.pre[i] = __builtin_shufflevector(lo[i], hi[i], 0, 1, 2, 3);
            __builtin_amdgcn_sched_barrier(0);
        };

        // plane transform B^T d B of this lane's (tile, word): strip rawT -> V[qbuf]  (wino_dw.hip)
        auto transform = [&]() __attribute__((always_inline)) {
            f32x4 ya[4], yb[4];
#pragma unroll
            for (int cc = 0; cc < 4; ++cc) {
                const int co = ((cc & 1) * 9 + (cc >> 1)) * kCB;
                const f32x4 R0 = *reinterpret_cast<const f32x4*>(rawT + rdR0 + co);
                const f32x4 R1 = *reinterpret_cast<const f32x4*>(rawT + rdR1 + co);
                const f32x4 R2 = *reinterpret_cast<const f32x4*>(rawT + rdR2 + co);
                ya[cc] = pk_fma_s(R1, m1, R0);
                yb[cc] = pk_fma_s(R2, sg, R1);
            }
            float* Vq = Vb + qbuf * kPcV;
            const int xa = (2 * thalf) * 4, xb = (2 * thalf + 1) * 4;
            *reinterpret_cast<f32x4*>(Vq + pc_slot(xa + 0, ttile, tword)) = pk_fma_s(ya[2], m1, ya[0]);
            *reinterpret_cast<f32x4*>(Vq + pc_slot(xa + 1, ttile, tword)) = pk_add(ya[1], ya[2]);
            *reinterpret_cast<f32x4*>(Vq + pc_slot(xa + 2, ttile, tword)) = pk_fma_s(ya[1], m1, ya[2]);
            *reinterpret_cast<f32x4*>(Vq + pc_slot(xa + 3, ttile, tword)) = pk_fma_s(ya[3], m1, ya[1]);
            *reinterpret_cast<f32x4*>(Vq + pc_slot(xb + 0, ttile, tword)) = pk_fma_s(yb[2], m1, yb[0]);
            *reinterpret_cast<f32x4*>(Vq + pc_slot(xb + 1, ttile, tword)) = pk_add(yb[1], yb[2]);
            *reinterpret_cast<f32x4*>(Vq + pc_slot(xb + 2, ttile, tword)) = pk_fma_s(yb[1], m1, yb[2]);
            *reinterpret_cast<f32x4*>(Vq + pc_slot(xb + 3, ttile, tword)) = pk_fma_s(yb[3], m1, yb[1]);
        };
        int gi = 0;                            // iterations so far (strip parity; the transform lags one iteration)

        for (int it = 0; it < count; ++it) {
            has_next = it + 1 < count;
            const bool interior = tl.y0 >= 1 && tl.y0 + kPcTH + 1 <= a.H && tl.x0 >= 1 && tl.x0 + kPcTW + 1 <= a.W;
            int cbi = 0, p = 0;
            for (int s = 0; s < NS; ++s) {
                // the book of the next tile is needed by the refills of the tile's last stage
                if (s == NS - 1 && has_next) { tn = d4_decode(first + (it + 1) * step, a); setup(tn, nxt_off, nxt_keep); }
                raw = rawb + (gi & 1) * kD4ShStrip; rawT = rawb + ((gi & 1) ^ 1) * kD4ShStrip;
                const int t = d4_t(p), cb = d4_cb(p, cbi, ncb), nsl = d4_nslot(t);
                // stage s + 1: (position, channel block, depth index), possibly of the next tile
                const bool nx = s + 1 >= NS;
                const int cbn = cbi + 1 == ncb ? 0 : cbi + 1, pn = nx ? 0 : (cbi + 1 == ncb ? p + 1 : p);
                const int tnx = d4_t(pn), cbne = d4_cb(pn, cbn, ncb);
                if constexpr (!IDENT) {
                    ssw[0] = *reinterpret_cast<const f32x4*>(ssl + 2 * (cb * kCB + w4 * 4));
                    ssw[1] = *reinterpret_cast<const f32x4*>(ssl + 2 * (cb * kCB + w4 * 4) + 4);
                }
                const int nsn = d4_nslot(tnx);
                // ---- the stage's depth combination D_t = sum_k c_k act(x[z_k]) IN REGISTERS (all four unit sets were requested a stage ago),
                // one strip write per word: wino_dw.hip's publish-then-combine through the strip (a read-modify-write of the LDS words per
                // further unit) would cost 9 more LDS reads and 6-9 more writes per lane and stage here.  A slice outside the volume enters
                // with coefficient 0 (its clamped load is finite); the zero padding of the plane is one multiply of the combined words.
                {
                    float c[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const int z = tl.z0 - 1 + d4_j(t, k < nsl ? k : 0);
                        c[k] = (k < nsl && z >= 0 && z < a.N) ? d4_c(t, k) : 0.f;
                    }
                    // unit by unit: wait for ITS words only, fold them into the running combination, request the same slot of stage s + 1
                    // right away (the refills stay spread over the stage: twelve loads in one burst cost ~600 issue cycles in a row, and one
                    // wait for all four sets exposes the slowest — measured +3 % against this order)
                    f32x2 lo[kD4NPF], hi[kD4NPF];
                    activate(set0);
                    {
                        // the first slot's coefficient is 1 (every row of Bd has one) unless its slice is outside the volume (then 0)
                        if (c[0] != 0.f) {
#pragma unroll
                            for (int i = 0; i < kD4NPF; ++i) { lo[i] = set0.pre[i].lo; hi[i] = set0.pre[i].hi; }
                        } else {
#pragma unroll
                            for (int i = 0; i < kD4NPF; ++i) { lo[i] = f32x2{0.f, 0.f}; hi[i] = f32x2{0.f, 0.f}; }
                        }
                    }
                    issue(nx && has_next, d4_j(tnx, 0), cbne, set0);
                    activate(set1);
                    {
                        const f32x2 c1 = {c[1], c[1]};
#pragma unroll
                        for (int i = 0; i < kD4NPF; ++i) { lo[i] = __builtin_elementwise_fma(set1.pre[i].lo, c1, lo[i]); hi[i] = __builtin_elementwise_fma(set1.pre[i].hi, c1, hi[i]); }
                    }
                    issue(nx && has_next, d4_j(tnx, 1), cbne, set1);
                    activate(set2);
                    {
                        const f32x2 c2 = {c[2], c[2]};
#pragma unroll
                        for (int i = 0; i < kD4NPF; ++i) { lo[i] = __builtin_elementwise_fma(set2.pre[i].lo, c2, lo[i]); hi[i] = __builtin_elementwise_fma(set2.pre[i].hi, c2, hi[i]); }
                    }
                    issue(nx && has_next, d4_j(tnx, 2), cbne, set2);
                    if (nsl > 3) {
                        activate(set3);
                        const f32x2 c3 = {c[3], c[3]};
#pragma unroll
                        for (int i = 0; i < kD4NPF; ++i) { lo[i] = __builtin_elementwise_fma(set3.pre[i].lo, c3, lo[i]); hi[i] = __builtin_elementwise_fma(set3.pre[i].hi, c3, hi[i]); }
                    }
                    if (nsn > 3) issue(nx && has_next, d4_j(tnx, 3), cbne, set3);
#pragma unroll
                    for (int i = 0; i < kD4NPF; ++i) {
                        if (!interior) {
                            const f32x2 kk = {cur_keep[i], cur_keep[i]};
                            lo[i] = lo[i] * kk; hi[i] = hi[i] * kk;
                        }
                        *reinterpret_cast<f32x4*>(raw + wr_off[i]) = __builtin_shufflevector(lo[i], hi[i], 0, 1, 2, 3);
                    }
                }
                // plane transform of the stage published one iteration ago
                if (gi > 0) transform();
                __syncthreads();
                if (gi > 0) qbuf ^= 1;
                ++gi;
                if (++cbi == ncb) { cbi = 0; ++p; }
            }
            tl = tn;
#pragma unroll
            for (int u = 0; u < kD4NPF; ++u) { cur_off[u] = nxt_off[u]; cur_keep[u] = nxt_keep[u]; }
        }
        {                                      // the last published stage
            rawT = rawb + ((gi & 1) ^ 1) * kD4ShStrip;
            transform();
            __syncthreads();
        }
        __syncthreads();                       // the consumers' last stage
    }
}

// w [Cout][Cin][3][3][3] -> U_t = sum_kd Gd[t][kd] (G g_kd G^T) (float64, rounded once) in the kernel's B-operand order, phases in
// EXECUTION order: [cg][stage = p*ncb + cb][xi][wave][lane = kq*16 + j][e], t = d4_t(p), co = cg*64 + 16*wave + j, ci = cb*16 + 4*kq + e
// transposed = 1: the data-gradient stream (w is stored [Cin][Cout][3][3][3] seen from this kernel: its ci is the stored tensor's output
// channel; taps flipped in every dimension); 2: both streams in one launch (grid.y = 2), the data gradient's behind the forward one.
__global__ __launch_bounds__(256) void conv_wino_dw4_pack_kernel(const float* __restrict__ w, float* __restrict__ wp, int Cin, int Cout,
                                                                 int transposed) {
    const long total = (long)Cout * Cin * 6 * 16;
    if (transposed == 2) {
        transposed = blockIdx.y;
        if (transposed) { const int c = Cin; Cin = Cout; Cout = c; wp += total; }
    }
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    long t = idx;
    const int e = t & 3; t >>= 2;
    const int j = t & 15; t >>= 4;
    const int kq = t & 3; t >>= 2;
    const int wave = t & 3; t >>= 2;
    const int xi = t & 15; t >>= 4;
    const int ncb = Cin / kCB;
    const int stage = (int)(t % (6 * ncb));
    const int cg = (int)(t / (6 * ncb));
    const int p = stage / ncb, cb = stage - p * ncb;
    const int td = p < 4 ? p + 1 : (p == 4 ? 0 : 5);
    const int co = cg * 64 + 16 * wave + j, ci = cb * kCB + 4 * kq + e;
    const double G[4][3] = {{1.0, 0.0, 0.0}, {0.5, 0.5, 0.5}, {0.5, -0.5, 0.5}, {0.0, 0.0, 1.0}};
    const double Gd[6][3] = {{16.0 / 9.0, 0.0, 0.0}, {-1.0, -0.5, -0.25}, {-1.0, 0.5, -0.25}, {1.0 / 9.0, 1.0 / 6.0, 0.25}, {1.0 / 9.0, -1.0 / 6.0, 0.25}, {0.0, 0.0, 1.0}};
    const int aa = xi >> 2, bb = xi & 3;
    double u = 0.0;
#pragma unroll
    for (int kd = 0; kd < 3; ++kd) {
        const float* g = transposed ? w + (((size_t)ci * Cout + co) * 3 + (2 - kd)) * 9 : w + (((size_t)co * Cin + ci) * 3 + kd) * 9;
        double u2 = 0.0;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) u2 += G[aa][ky] * (double)g[transposed ? (2 - ky) * 3 + (2 - kx) : ky * 3 + kx] * G[bb][kx];
        u += Gd[td][kd] * u2;
    }
    wp[idx] = (float)u;
}

}  // namespace nrgbd

extern "C" int nrgbd_conv_wino_dw4_pack(const float* w, float* w_wino, int Cin, int Cout, int transposed, void* stream) {
    using namespace nrgbd;
    if (!w || !w_wino) return NRGBD_E_NULL;
    if (Cin <= 0 || Cin % kCB || Cout <= 0 || Cout % 64) return NRGBD_E_SHAPE;
    if (transposed < 0 || transposed > 2) return NRGBD_E_ARG;
    if (transposed == 2 && Cin % 64) return NRGBD_E_SHAPE;
    const long total = (long)Cout * Cin * 6 * 16;
    hipLaunchKernelGGL(conv_wino_dw4_pack_kernel, dim3((unsigned)((total + 255) / 256), transposed == 2 ? 2 : 1), dim3(256), 0,
                       (hipStream_t)stream, w, w_wino, Cin, Cout, transposed);
    NRGBD_CHECK_LAUNCH();
    return NRGBD_OK;
}

static int dw4_workgroups(int N, int H, int W, int Cout, int* out) {
    const long nt = (long)(nrgbd_conv_wino_tiles(N, H, W, 1) / 4) * (Cout / 64);
    int dev = 0, ncu = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e == hipSuccess) e = hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev);
    if (e != hipSuccess) return (int)e;
    if (ncu <= 0) return NRGBD_E_ARG;
    *out = nt < ncu ? (int)nt : ncu;
    return NRGBD_OK;
}

extern "C" int nrgbd_conv_wino_dw4_workspace(int N, int H, int W, int Cout, size_t* bytes) {
    if (!bytes) return NRGBD_E_NULL;
    if (N <= 0 || (N & 3) || H <= 0 || W <= 0 || Cout <= 0 || Cout % 64) return NRGBD_E_SHAPE;
    int n = 0;
    const int rc = dw4_workgroups(N, H, W, Cout, &n);
    if (rc != NRGBD_OK) return rc;
    *bytes = (size_t)n * 4 * nrgbd::kD4ScratchWave * sizeof(float);
    return NRGBD_OK;
}

extern "C" int nrgbd_conv_wino_dw4_f32(const float* x, const float* x_ss, int x_relu, float x_unit, const float* w_wino, float* y,
                                       float* stats, void* workspace, size_t workspace_bytes, int N, int H, int W, int Cin, int Cout,
                                       void* stream) {
    using namespace nrgbd;
    if (!x || !w_wino || !y || !workspace) return NRGBD_E_NULL;
    if (N <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cin % kCB || Cin > kD4MaxCin || Cout <= 0 || Cout % 64) return NRGBD_E_SHAPE;
    if (N & 3) return NRGBD_E_SHAPE;                                  // quadruples of output slices
    if (H % kPcTH || W % kPcTW) return NRGBD_E_SHAPE;                 // whole 8x16 tiles only
    if ((long)H * W * Cin >= (1L << 30)) return NRGBD_E_SHAPE;       // 32-bit BYTE offsets inside a slice
    if (reinterpret_cast<uintptr_t>(workspace) & 15) return NRGBD_E_ALIGN;
    const bool clamp = x_unit != 0.f;
    if (clamp) {
        int ex = 0;
        if (!x_ss || !x_relu || !(x_unit > 0.f) || x_unit > 1.f || frexpf(x_unit, &ex) != 0.5f) return NRGBD_E_ARG;   // a power of two in (0, 1]
    }
    const int rows = nrgbd_conv_wino_tiles(N, H, W, 1);              // statistics rows: one per (8x16 tile, slice)
    const long nt = (long)(rows / 4) * (Cout / 64);
    if (nt >= (1L << 31)) return NRGBD_E_SHAPE;
    int nwg = 0;
    const int rc = dw4_workgroups(N, H, W, Cout, &nwg);
    if (rc != NRGBD_OK) return rc;
    if (workspace_bytes < (size_t)nwg * 4 * kD4ScratchWave * sizeof(float)) return NRGBD_E_NULL;
    WinoD4Args aa{};
    aa.b = WinoPcArgs{x, x_ss, nullptr, nullptr, nullptr, w_wino, y, stats, x_relu, 0, N, H, W, Cin, Cout, (int)nt, rows,
                      nullptr, 0, 0, 0, 0, 0, x_unit};
    aa.scratch = static_cast<float*>(workspace);
    const size_t lds = (size_t)(kD4NBuf * kPcV + 2 * kD4ShStrip + 4 * kD4StashWave + 2 * Cin) * sizeof(float);
    // the function's opt-in is set to the form's maximum, not to this call's size (see nrgbd_conv_wino_f32: hipGraph replays read it)
    const int lds_attr = (int)((size_t)(kD4NBuf * kPcV + 2 * kD4ShStrip + 4 * kD4StashWave + 2 * kD4MaxCin) * sizeof(float));
    hipStream_t st = (hipStream_t)stream;
    hipError_t e;
#define NRGBD_D4_LAUNCH(ID_, CL_)                                                                                          \
    do {                                                                                                                   \
        e = set_max_dynamic_lds(reinterpret_cast<const void*>(&conv_wino_dw4_kernel<ID_, CL_>),                            \
                                lds_attr);                                     \
        if (e != hipSuccess) return (int)e;                                                                                \
        hipLaunchKernelGGL((conv_wino_dw4_kernel<ID_, CL_>), dim3(nwg), dim3(512), lds, st, aa);                           \
    } while (0)
    if (clamp) NRGBD_D4_LAUNCH(false, true);
    else if (!x_ss && !x_relu) NRGBD_D4_LAUNCH(true, false);
    else NRGBD_D4_LAUNCH(false, false);
#undef NRGBD_D4_LAUNCH
    NRGBD_CHECK_LAUNCH();
    return NRGBD_OK;
}
