// wino_pc.hip — 3x3(x3) convolutions in the Winograd domain F(2x2, 3x3) on the fp32 matrix cores of gfx950, second generation:
// a PERSISTENT workgroup of 8 waves split into 4 consumer waves that do nothing but issue MFMAs and 4 producer waves that
// load, normalise, transform and publish the operand of the stage two steps ahead.
//
// Serves  (a) the K-Net's ten 64 -> 64 3x3x3 layers (models/basic.py:71-94): KD = 3 depth taps, each a 2-D Winograd problem;
//         (b) the 3x3 stride-1 layers of the feature CNN (models/psm_submodule.py:10-16,31-50,100-134), dilation 1 or 2:
//             KD = 1, any Cin % 16 == 0, Cout % 64 == 0 (a tile is repeated per 64-column group of outputs)
// with the same fused BatchNorm work around them as conv3d.hip / conv2d.hip: statistics of the raw output in the epilogue,
// normalise + ReLU + residual add (+ materialise) while the input is loaded.
//
// Why a second generation.  conv3d_wino.hip (generation 1) runs publish -> barrier -> transform -> barrier -> 128 MFMAs per
// stage in every wave: at 2 workgroups per CU the matrix pipe idles whenever both resident waves of a SIMD are in their
// load / transform phases or wait for the first weight line after the barrier (measured: 3.36 ms per layer at the
// 192x256x64 grid = 59 % of the Winograd-domain MFMA time).  Here the two kinds of work live in different waves:
//   consumer wave c (waves 4..7) = output channels 16c .. 16c+15 of ALL 16 transform points and all 32 tiles of the workgroup's
//       8x16-pixel tile (128 accumulator VGPRs): per stage 16 x (2 LDS reads + 1 weight line + 8 v_mfma_f32_16x16x4_f32);
//       weight lines (1 KB, packed per wave) run 7 steps ahead in an 8-deep register ring that continues across stages and
//       tiles; the first A operand of the next stage is read before the stage barrier (three V buffers make that legal);
//       the first k-step of a tile takes a zero C operand (the accumulators are never cleared).
//   producer waves (waves 0..3: the older waves win the SIMD's VALU arbitration): the 10x18 halo of one 16-channel block is
//       split over their 256 lanes (3 16-byte words each: every halo word has ONE loader — round 4; until then wave p loaded the
//       four rows 2p..2p+3 of its tile row into a private strip, 16 rows for a 10-row halo): global -> registers (buffer loads,
//       TWO stages ahead, two register sets, the stage's (scale, shift) with them) -> BatchNorm / ReLU / residual, packed and
//       breadth-first -> the SHARED strip of the stage (two of them alternate); the transform B^T d B per (tile, 16-byte word,
//       half) -> V[q % 3] of producer wave p = tile row p (8 Winograd tiles) runs one iteration LATER, on the strip the stage
//       barrier has completed — no other synchronisation.
//   One s_barrier per stage; the consumers never wait for data (producers are two stages ahead), the producers wait for
//   the consumers — which is the point: the matrix pipe is the resource to keep busy.  What was measured on the way
//   (in-kernel clocks, profiles/r2_pmc_wino.txt): beside a wave that streams MFMAs a partner's VALU
//   instruction issues about once per MFMA, a dependent one misses its slot, so the producers' code is written for
//   instruction count and independence, not for FLOPs.
//   Persistent: one workgroup per CU walks its share of the tile list (XCD-aware: an XCD's workgroups sweep neighbouring
//   tiles, depth fastest, so the three slices a 3-D tile needs are shared in that XCD's L2), so a tile's epilogue and the
//   next tile's first loads overlap with the producers' run-ahead instead of being exposed at every workgroup boundary.
// LDS: 3 x 32 KB V + 2 x 12.8 KB strips = 122 KB (one workgroup per CU; 2 waves per SIMD, up to 256 VGPRs each).
#include "wino_pc.hpp"

namespace nrgbd {

// MAT: the activated input is also written out (a.mat).  A template parameter because of what its stores do to the variants
// WITHOUT them (round 3, learnt on wino_dw.hip): once loads and stores of one wave can both be pending the
// compiler turns every s_waitcnt on a prefetched register into vmcnt(0), which also waits for the refill issued a moment
// earlier.  For the same reason the refills are unconditional and the (scale, shift) pairs come from an LDS copy.
// HALF: Cout = 32 (the feature CNN's half-resolution layers, psm_submodule.py:90-99,103: firstconv.1/.2 and layer1).  The four
// consumer waves split the tile as (row block m = wave >> 1: 16 of the 32 Winograd tiles) x (16-column group = wave & 1): 64
// accumulators and 64 MFMAs per stage and wave, transform points taken in PAIRS so that MFMAs on one accumulator still
// alternate with another's; the weight stream is the 64-column one with the upper 32 columns zero (waves 0 / 1 read their two
// lines of it); statistics rows are (tile, row block).  The producers are unchanged — they now set the pace (a stage's MFMAs
// take 0.85 us): 2.25x fewer multiplies than conv2d.hip's direct form of these layers.
template <int KD, int DIL, bool RES, bool ODD = false, int EPI = 0, bool MAT = false, bool HALF = false>
__global__ __launch_bounds__(512) void conv_wino_pc_kernel(const WinoPcArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* Vb = lds;                           // [3][16 xi][32 tiles][16]
    float* rawb = lds + kPcNBuf * kPcV;        // SHARED: [2][10 rows][20 pixels][16]; else [4 producer waves][4 rows][20 pixels][16]
    float* ssl = rawb + kPcStrips;             // [Cin][2] (scale, shift) of x, then [Cin][2] of res (identity where null)

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wv = wave & 3;                   // index within the role
    const int NS = (a.Cin / kCB) * KD;         // stages per tile
#ifdef NRGBD_DEV
    const int abl = a.abl;
    const long t_entry = wall_clock64();
#else
    constexpr int abl = 0;
#endif

    // ---- this workgroup's share of the tile list: XCD x = blockIdx % 8 owns the x-th contiguous eighth, its workgroups
    //      (slots) walk it interleaved, so the workgroups of one XCD are always on neighbouring tiles
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
    if (first >= end) return;                  // uniform: no wave of this workgroup ever reaches a barrier
    const int count = (end - first + step - 1) / step;
    const unsigned plane = (unsigned)((size_t)a.H * a.W * a.Cin);
    if constexpr (EPI == 0) {
        for (int i = threadIdx.x; i < 2 * a.Cin; i += 512) {
            const int j = pc_ss_slot(i);  // pairs as the packed FMAs take them: (s0, s1, t0, t1 | s2, s3, t2, t3) per 4 channels
            ssl[j] = a.x_ss ? a.x_ss[i] : ((i & 1) ? 0.f : 1.f);
            ssl[2 * a.Cin + j] = (RES && a.res_ss) ? a.res_ss[i] : ((i & 1) ? 0.f : 1.f);
        }
        __syncthreads();
    }

    // Producers are waves 0-3: VALU issue on a SIMD is arbitrated by age, and the producers' (few) VALU instructions have to
    // get through beside the consumer's continuous MFMA stream (measured: 3.58 -> 3.38 ms per layer from this alone)
    if (wave >= 4) {
        // =========================================== consumer: 16 output channels x 16 xi x 32 tiles ====================
        const int kq = lane >> 4, jj = lane & 15;
        constexpr int NM = HALF ? 1 : 2;       // row blocks (16 Winograd tiles each) of this wave
        const int msel = HALF ? (wv >> 1) : 0; // HALF: the one row block this wave owns
        const int cwv = HALF ? (wv & 1) : wv;  // the wave's 16-column group
        f32x4 acc[16][NM];                     // written by the first stage of every tile (C operand = 0): never cleared
        const int a0 = pc_slot(0, 16 * msel + jj, kq), a1 = pc_slot(0, 16 + jj, kq);   // + xi * 512 floats + buffer
        const f32x4* wbase = reinterpret_cast<const f32x4*>(a.wp) + cwv * 64 + lane;
        const unsigned ldy = (EPI == 1 && a.ldy) ? (unsigned)a.ldy : (unsigned)a.Cout;   // pixel stride of the output
        const unsigned lane_yoff = (unsigned)jj + (unsigned)((DIL * 2 * (kq >> 1)) * a.W + 8 * (kq & 1) * DIL) * ldy;
        const size_t wgroup = (size_t)NS * 16 * 256;                        // f32x4 per 64-column output group

        PcTile tl = pc_decode<KD, DIL>(first, a);
        const f32x4* wt = wbase + (size_t)tl.cg * wgroup;
        constexpr int BD = HALF ? 6 : kPcBD;   // weight lines in flight (HALF requests two per pair of points: an even distance)
        f32x4 Bn[kPcNB], An[4][2];             // A operands run TWO transform points ahead (one point = 256 MFMA cycles < a loaded LDS's latency)
                                               // HALF: An[pair & 3][point of the pair], two PAIRS ahead
#pragma unroll
        for (int b = 0; b < BD; ++b) Bn[b] = wt[b * 256];
        if constexpr (NRGBD_PC_SHARED != 0) __syncthreads();   // the producers publish stage 0 (transformed one iteration later)
        __syncthreads();                       // producers finish stage 0
        __syncthreads();                       // ... and stage 1
        if constexpr (HALF) {
            An[0][0] = *reinterpret_cast<const f32x4*>(Vb + a0);
            An[0][1] = *reinterpret_cast<const f32x4*>(Vb + a0 + kPcTiles * kCB);
            An[1][0] = *reinterpret_cast<const f32x4*>(Vb + a0 + 2 * kPcTiles * kCB);
            An[1][1] = *reinterpret_cast<const f32x4*>(Vb + a0 + 3 * kPcTiles * kCB);
        } else {
            An[0][0] = *reinterpret_cast<const f32x4*>(Vb + a0);
            An[0][1] = *reinterpret_cast<const f32x4*>(Vb + a1);
            An[1][0] = *reinterpret_cast<const f32x4*>(Vb + a0 + kPcTiles * kCB);
            An[1][1] = *reinterpret_cast<const f32x4*>(Vb + a1 + kPcTiles * kCB);
        }
        int buf = 0;
#ifdef NRGBD_DEV
        long t_mfma = 0, t_bar = 0, t_epi = 0;
        const long t_loop = wall_clock64();
#endif
        for (int it = 0; it < count; ++it) {
            const int tnext = first + (it + 1 < count ? it + 1 : it) * step;
            const PcTile tn = pc_decode<KD, DIL>(tnext, a);
            const f32x4* wt_next = wbase + (size_t)tn.cg * wgroup;
            for (int s = 0; s < NS; ++s) {
                const float* Vc = Vb + buf * kPcV;
                const int nbuf = buf == kPcNBuf - 1 ? 0 : buf + 1;
                const float* Vn = Vb + nbuf * kPcV;
                const f32x4* wcur = wt + (size_t)s * (16 * 256);
                const f32x4* wnx = s + 1 < NS ? wcur + 16 * 256 : wt_next;
#ifdef NRGBD_DEV
                const long c0 = wall_clock64();
#endif
                // one stage = 16 transform points x (2 A reads + 1 weight line + 8 MFMAs).  FIRST (stage 0 of a tile): the first
                // k-step takes a zero C operand instead of the accumulator, which saves clearing 128 registers per tile.
                auto body = [&](auto first_tag) __attribute__((always_inline)) {
                    constexpr bool FIRST = decltype(first_tag)::value;
                    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
                    if constexpr (HALF) {
                        // a step = the PAIR of transform points (2 xp, 2 xp + 1): 2 A reads (two pairs ahead), 8 MFMAs alternating
                        // between the two accumulators, the weight lines of points 2 xp + 6 and 2 xp + 7 requested in two MFMA gaps
#pragma unroll
                        for (int xp = 0; xp < 8; ++xp) {
                            const int cur = xp & 3, nxt = (xp + 2) & 3;
                            const float* Vs = xp + 2 < 8 ? Vc : Vn;   // pairs 0, 1 of the next stage: its buffer was completed two barriers ago
                            const int pn = (xp + 2) & 7;
                            An[nxt][0] = *reinterpret_cast<const f32x4*>(Vs + a0 + (2 * pn) * (kPcTiles * kCB));
                            An[nxt][1] = *reinterpret_cast<const f32x4*>(Vs + a0 + (2 * pn + 1) * (kPcTiles * kCB));
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                acc[2 * xp][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(An[cur][0][e], Bn[(2 * xp) % kPcNB][e],
                                                                                      FIRST && e == 0 ? zero4 : acc[2 * xp][0], 0, 0, 0);
                                acc[2 * xp + 1][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(An[cur][1][e], Bn[(2 * xp + 1) % kPcNB][e],
                                                                                          FIRST && e == 0 ? zero4 : acc[2 * xp + 1][0], 0, 0, 0);
                                if (e == 1) Bn[(2 * xp + 6) % kPcNB] = 2 * xp + 6 < 16 ? wcur[(2 * xp + 6) * 256] : wnx[(2 * xp + 6 - 16) * 256];
                                if (e == 3) Bn[(2 * xp + 7) % kPcNB] = 2 * xp + 7 < 16 ? wcur[(2 * xp + 7) * 256] : wnx[(2 * xp + 7 - 16) * 256];
                                __builtin_amdgcn_sched_barrier(0);
                            }
                        }
                    } else {
#pragma unroll
                    for (int xi = 0; xi < 16; ++xi) {
                        const int cur = xi & 3, nxt = (xi + 2) & 3;
                        if (xi + 2 < 16) {
                            An[nxt][0] = *reinterpret_cast<const f32x4*>(Vc + a0 + (xi + 2) * (kPcTiles * kCB));
                            An[nxt][1] = *reinterpret_cast<const f32x4*>(Vc + a1 + (xi + 2) * (kPcTiles * kCB));
                        } else {   // the first two operands of the next stage: its buffer was completed two barriers ago
                            An[nxt][0] = *reinterpret_cast<const f32x4*>(Vn + a0 + (xi + 2 - 16) * (kPcTiles * kCB));
                            An[nxt][1] = *reinterpret_cast<const f32x4*>(Vn + a1 + (xi + 2 - 16) * (kPcTiles * kCB));
                        }
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            acc[xi][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(An[cur][0][e], Bn[xi % kPcNB][e],
                                                                              FIRST && e == 0 ? zero4 : acc[xi][0], 0, 0, 0);
                            acc[xi][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(An[cur][1][e], Bn[xi % kPcNB][e],
                                                                              FIRST && e == 0 ? zero4 : acc[xi][1], 0, 0, 0);
                            // pin the order: the two row blocks alternate (no back-to-back dependent MFMAs) and the operand
                            // streams keep their distances
                            // the weight line of the point 7 ahead is requested HERE, in the second MFMA gap of the point, not at its top beside the two
                            // LDS reads: a vector-memory instruction costs the wave ~50 issue cycles, and three memory instructions in one gap let the
                            // matrix pipe run dry (tools/probes/mfma_stream_probe.hip: 78.5 -> 85.4 % busy)
                            if (e == NRGBD_WPOS) Bn[(xi + kPcBD) % kPcNB] = xi + kPcBD < 16 ? wcur[(xi + kPcBD) * 256] : wnx[(xi + kPcBD - 16) * 256];
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    }
                    }
                };
                if (!(abl & 1)) {
                    if (s == 0) body(std::true_type{}); else body(std::false_type{});
                }
#ifdef NRGBD_DEV
                const long c1 = wall_clock64();
#endif
                __syncthreads();
#ifdef NRGBD_DEV
                const long c2 = wall_clock64();
                t_mfma += c1 - c0; t_bar += c2 - c1;
#endif
                buf = nbuf;
            }
#ifdef NRGBD_DEV
            const long c3 = wall_clock64();
#endif
            // ---- inverse transform Y = A^T M A in registers + output + per-channel partial statistics ----
            // lane (kq, jj): output channel co = 16 wv + jj; register r of row block m = tile 16 m + 4 kq + r, i.e. tile row
            // 2m + (kq >> 1), tile column 4 (kq & 1) + r: the lane part of an output's address is loop-invariant (lane_yoff),
            // the (m, r, a) part is uniform -> scalar base + 32-bit lane offset stores, no per-store address arithmetic
            const int co = tl.cg * 64 + cwv * 16 + jj;
            float* ybase = a.y + (((size_t)tl.n * a.H + tl.y0 + tl.py) * a.W + tl.x0 + tl.px) * ldy + (EPI == 1 ? a.ycoff : 0) + tl.cg * 64 + cwv * 16;
            const bool cok = EPI != 1 || a.cout_valid == 0 || co < a.cout_valid;   // EPI = 1: a padded output column is not stored
            const bool inside = tl.y0 + tl.py + DIL * (kPcTH - 1) < a.H && tl.x0 + tl.px + DIL * (kPcTW - 1) < a.W;
            float s1 = 0.f, s2 = 0.f;
            float bval = 0.f;
            if constexpr (EPI == 1) bval = a.bias ? a.bias[co] : 0.f;
            if (inside) {
                // interior tile: the whole inverse transform, the statistics and the stores on register PAIRS (tiles r, r+1 of a
                // row block): v_pk_add_f32 / v_pk_fma_f32 halve the epilogue's VALU instructions; a - b is fma(b, -1, a) with
                // an opaque -1 (same rounding; a literal would be folded into two scalar v_sub)
                float neg1 = -1.f;
                asm volatile("" : "+v"(neg1));
                const f32x2 n1 = {neg1, neg1};
                const f32x2 bias2 = {bval, bval};
                f32x2 S1 = {0.f, 0.f}, S2 = {0.f, 0.f};
#pragma unroll
                for (int mi = 0; mi < NM; ++mi) {
                    const int m = HALF ? msel : mi;   // row block: address arithmetic uses m, the register index is mi
#pragma unroll
                    for (int rp = 0; rp < 2; ++rp) {
                        f32x2 tr[2][4];   // t[a][xi_x] = sum_xi_y A^T[a][xi_y] M[xi_y][xi_x]
#pragma unroll
                        for (int xx = 0; xx < 4; ++xx) {
                            const f32x2 m0 = rp ? acc[0 + xx][mi].hi : acc[0 + xx][mi].lo, m1 = rp ? acc[4 + xx][mi].hi : acc[4 + xx][mi].lo;
                            const f32x2 m2 = rp ? acc[8 + xx][mi].hi : acc[8 + xx][mi].lo, m3 = rp ? acc[12 + xx][mi].hi : acc[12 + xx][mi].lo;
                            tr[0][xx] = (m0 + m1) + m2;
                            tr[1][xx] = __builtin_elementwise_fma(m3, n1, __builtin_elementwise_fma(m2, n1, m1));   // (m1 - m2) - m3
                        }
#pragma unroll
                        for (int aa = 0; aa < 2; ++aa) {
                            f32x2 o0 = (tr[aa][0] + tr[aa][1]) + tr[aa][2];
                            f32x2 o1 = __builtin_elementwise_fma(tr[aa][3], n1, __builtin_elementwise_fma(tr[aa][2], n1, tr[aa][1]));
                            if constexpr (EPI == 1) {   // m_submodule.py:18-27: bias, LeakyReLU(0.01) = max(z, 0.01 z)
                                o0 = o0 + bias2; o1 = o1 + bias2;
                                if (a.out_lrelu) {
                                    const f32x2 sl = {0.01f, 0.01f};
                                    o0 = __builtin_elementwise_max(o0, o0 * sl); o1 = __builtin_elementwise_max(o1, o1 * sl);
                                }
                            }
                            float* oa = ybase + ((size_t)(DIL * (4 * m + aa)) * a.W + (size_t)(2 * (2 * rp) * DIL)) * ldy;       // tile r = 2 rp
                            float* ob = ybase + ((size_t)(DIL * (4 * m + aa)) * a.W + (size_t)(2 * (2 * rp + 1) * DIL)) * ldy;   // tile r + 1
                            if (cok) {
                                oa[lane_yoff] = o0.x; oa[lane_yoff + DIL * ldy] = o1.x;
                                ob[lane_yoff] = o0.y; ob[lane_yoff + DIL * ldy] = o1.y;
                            }
                            S1 = (S1 + o0) + o1;
                            S2 = __builtin_elementwise_fma(o1, o1, __builtin_elementwise_fma(o0, o0, S2));
                        }
                    }
                }
                s1 = S1.x + S1.y; s2 = S2.x + S2.y;
            } else {
#pragma unroll
                for (int mi = 0; mi < NM; ++mi) {
                    const int m = HALF ? msel : mi;   // row block: address arithmetic uses m, the register index is mi
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float tr[2][4];
#pragma unroll
                        for (int xx = 0; xx < 4; ++xx) {
                            const float m0 = acc[0 + xx][mi][r], m1 = acc[4 + xx][mi][r], m2 = acc[8 + xx][mi][r], m3 = acc[12 + xx][mi][r];
                            tr[0][xx] = (m0 + m1) + m2;
                            tr[1][xx] = (m1 - m2) - m3;
                        }
#pragma unroll
                        for (int aa = 0; aa < 2; ++aa) {
                            float o0 = (tr[aa][0] + tr[aa][1]) + tr[aa][2];
                            float o1 = (tr[aa][1] - tr[aa][2]) - tr[aa][3];
                            if constexpr (EPI == 1) {
                                o0 += bval; o1 += bval;
                                if (a.out_lrelu) { o0 = fmaxf(o0, 0.01f * o0); o1 = fmaxf(o1, 0.01f * o1); }
                            }
                            float* o = ybase + ((size_t)(DIL * (4 * m + aa)) * a.W + (size_t)(2 * r * DIL)) * ldy;   // uniform
                            const int tile = 16 * m + 4 * kq + r;
                            const int gy = tl.y0 + tl.py + DIL * (2 * (tile >> 3) + aa), gx = tl.x0 + tl.px + DIL * (2 * (tile & 7));
                            if (gy < a.H && cok) {
                                if (gx < a.W) { o[lane_yoff] = o0; s1 += o0; s2 = __builtin_fmaf(o0, o0, s2); }
                                if (gx + DIL < a.W) { o[lane_yoff + DIL * ldy] = o1; s1 += o1; s2 = __builtin_fmaf(o1, o1, s2); }
                            }
                        }
                    }
                }
            }
            if (a.stats) {   // the wave owns its 16 channels: reduce over the 4 lanes (kq) that share a channel
                s1 += __shfl_xor(s1, 16, 64); s2 += __shfl_xor(s2, 16, 64);
                s1 += __shfl_xor(s1, 32, 64); s2 += __shfl_xor(s2, 32, 64);
                if (kq == 0) {
                    // column-major partials [2 Cout][rows]: nrgbd_bn_finalize_cm reads a channel's partials as one run
                    const int srow = HALF ? 2 * tl.row + msel : tl.row;   // HALF: two waves share a channel -> a row per (tile, row block)
                    a.stats[(size_t)co * a.rows + srow] = s1;
                    a.stats[(size_t)(a.Cout + co) * a.rows + srow] = s2;
                }
            }
            tl = tn;
            wt = wt_next;
#ifdef NRGBD_DEV
            t_epi += wall_clock64() - c3;
#endif
        }
#ifdef NRGBD_DEV
        if ((abl & 64) && a.stats && wv == 0 && lane == 0) {   // timing record over the first statistics rows (results invalid)
            float* o = a.stats + (size_t)blockIdx.x * 8;
            o[0] = (float)t_mfma; o[1] = (float)t_bar; o[2] = (float)t_epi; o[3] = (float)count;
            float* o3 = a.stats + 8192 + (size_t)blockIdx.x * 4;   // absolute 100 MHz ticks (low 24 bits: exact in a float)
            o3[0] = (float)(t_entry & 0xFFFFFF); o3[1] = (float)(t_loop & 0xFFFFFF); o3[2] = (float)(wall_clock64() & 0xFFFFFF); o3[3] = 0.f;
        }
#endif
    } else {
        // =========================================== producer: tile row pw (8 Winograd tiles) ===========================
        const int pw = wv;
        constexpr bool SH = NRGBD_PC_SHARED != 0;
        constexpr int kItems = SH ? kPcShItems : kPcItems;
        float* raw = SH ? rawb : rawb + pw * kPcRawWave;   // SH: the strip this iteration PUBLISHES into (set per iteration)
        const float* rawT = raw;                            // ... and the one it TRANSFORMS from
        const int w4 = lane & 3;
        // load / publish items: item = lane + 64u -> strip pixel pi = item >> 2 in (row, de-interleaved column) order
        // item u of this lane -> strip row, image-order column and strip offset (recomputed where needed: a few integer
        // operations instead of 15 live registers)
        // SH: item = 192 pw + lane + 64 u over the whole 10-row halo (every halo word has ONE loader)
        auto item_id = [&](int u) { return (SH ? 192 * pw : 0) + lane + 64 * u; };
        auto item_rr = [&](int u) { return (item_id(u) >> 2) / 18; };
        auto item_cp = [&](int u) { const int pi = item_id(u) >> 2; return pi - (pi / 18) * 18; };
        auto item_col = [&](int u) { const int cp = item_cp(u); return cp < 9 ? 2 * cp : 2 * cp - 17; };   // even columns first, then odd
        // strip offset an item is published at; the 32 lanes without a fifth item (288 = 4.5 x 64) write a zero into the
        // strip's 8 pad pixels (columns 18, 19 of the 20-pixel row pitch) so that the publish loop has no per-lane branch
        int wr_off[kPcNPFx];
#pragma unroll
        for (int u = 0; u < kPcNPFx; ++u) {
            const int item = item_id(u), e = (item - kItems) >> 2;
            wr_off[u] = item < kItems ? (item_rr(u) * kPcRawW + item_cp(u)) * kCB + w4 * 4
                                      : ((e >> 1) * kPcRawW + 18 + (e & 1)) * kCB + w4 * 4;
        }
        // transform item of this lane: (tile of the row, 16-byte word, half of the xi rows); 16 consecutive lanes = 4 tiles x 4
        // words: conflict-free strip reads (the four tiles' columns are consecutive strip pixels) and V writes
        const int tword = lane & 3, txl = ((lane >> 5) << 2) | ((lane >> 2) & 3), thalf = (lane >> 4) & 1;
        const int ttile = pw * 8 + txl;
        // Row transform without per-lane selects: the lane reads its three strip rows in a lane-dependent ORDER (R0, R1, R2) and
        // computes ya = R0 - R1, yb = R1 + sg * R2:
        //   half 0 (xi_y 0, 1): R = strip rows (0, 2, 1), sg = +1 ->  d0 - d2,  d2 + d1
        //   half 1 (xi_y 2, 3): R = strip rows (2, 1, 3), sg = -1 ->  d2 - d1,  d1 - d3
        // (strip columns of tile txl: cc = 0 at column pixel txl, cc = 1: +9 pixels, cc = 2: +1, cc = 3: +10)
        const int rdc = txl * kCB + tword * 4 + (SH ? 2 * pw * kPcRawW * kCB : 0);   // SH: the tile row's halo rows start at strip row 2 pw
        const int rdR0 = (thalf ? 2 : 0) * kPcRawW * kCB + rdc, rdR1 = (thalf ? 1 : 2) * kPcRawW * kCB + rdc,
                  rdR2 = (thalf ? 3 : 1) * kPcRawW * kCB + rdc;
        const float sg = thalf ? -1.f : 1.f;
        float m1 = -1.f;                    // opaque to the optimiser: fma(a, -1, c) would otherwise be folded into four scalar
        asm volatile("" : "+v"(m1));        // v_sub_f32; as a register operand it stays one v_pk_fma_f32 per pair (same rounding)

        // Per-tile bookkeeping of this lane's items: in-plane element offset (a harmless in-tensor offset when outside),
        // inside-the-image bits, owner bits (materialise target).  Two books: the raw words run TWO stages ahead of their
        // use, so the last two stages of a tile already load the next tile's words.
        unsigned cur_off[kPcNPFx], cur_own = 0, nxt_off[kPcNPFx], nxt_own = 0;   // BYTE offsets
        float cur_keep[kPcNPFx], nxt_keep[kPcNPFx];                               // 1 inside the image, 0 outside (zero padding)
        auto setup = [&](const PcTile& t, unsigned (&b_off)[kPcNPFx], float (&b_keep)[kPcNPFx], unsigned& b_own) __attribute__((always_inline)) {
            b_own = 0;
#pragma unroll
            for (int u = 0; u < kPcNPFx; ++u) {
                const int rr = item_rr(u), hy = SH ? rr : 2 * pw + rr, hx = item_col(u);
                const int gy = t.y0 + t.py + DIL * (hy - 1), gx = t.x0 + t.px + DIL * (hx - 1);
                const bool in = item_id(u) < kItems && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
                const unsigned n2 = KD == 3 ? 0u : (unsigned)t.n;
                b_off[u] = 4u * (in ? (unsigned)((((size_t)n2 * a.H + gy) * a.W + gx) * a.Cin + w4 * 4) : (unsigned)(w4 * 4));
                b_keep[u] = in ? 1.f : 0.f;
                const bool mine = SH ? (hy >= 1 && hy <= kPcTH) : (rr == 1 || rr == 2);   // SH: the one loader of a pixel of the tile's own 8 x 16
                if (in && mine && hx >= 1 && hx <= kPcTW) b_own |= 1u << u;
            }
        };
        // One register set per stage parity: raw words (+ residual words) and the (scale, shift) of the stage's 4 channels.
        // A set is refilled for stage s+2 right after stage s has published it, so a load has two stage periods to land:
        // with a single set the chain load -> publish -> next load made the producers' period = memory latency + publish
        // (measured 2.8 us against the consumers' 2.0 us of MFMAs), i.e. the matrix pipe waited for the producers.
        struct Regs { f32x4 pre[kPcNPFx]; f32x4 prer[RES ? kPcNPFx : 1]; f32x4 ss[2]; f32x4 rs[2]; };
        PcTile tl = pc_decode<KD, DIL>(first, a), tn = tl;
        // raw words of stage s -> registers; nx: the stage belongs to the NEXT tile (book nxt_*, tile tn).  The book is
        // selected per value, not per pointer: a pointer select would force both books into scratch memory.
        auto issue = [&](bool nx, int s, Regs& r) __attribute__((always_inline)) {
            const int cb = s / KD, kd = s - cb * KD;
            r.ss[0] = r.ss[1] = r.rs[0] = r.rs[1] = f32x4{1.f, 1.f, 0.f, 0.f};
            if constexpr (EPI == 0) {
                r.ss[0] = *reinterpret_cast<const f32x4*>(ssl + 2 * (cb * kCB + w4 * 4));
                r.ss[1] = *reinterpret_cast<const f32x4*>(ssl + 2 * (cb * kCB + w4 * 4) + 4);
                if constexpr (RES) {
                    r.rs[0] = *reinterpret_cast<const f32x4*>(ssl + 2 * a.Cin + 2 * (cb * kCB + w4 * 4));
                    r.rs[1] = *reinterpret_cast<const f32x4*>(ssl + 2 * a.Cin + 2 * (cb * kCB + w4 * 4) + 4);
                }
            }
            const int tz = nx ? tn.n : tl.n;
            const int z = KD == 3 ? min(max(tz + kd - 1, 0), a.N - 1) : 0;   // clamped: an outside slice is zeroed when published
            // uniform 64-bit base + per-lane 32-bit byte offset: the global_load saddr form, no per-lane 64-bit address math
            const size_t base = ((size_t)z * plane + (size_t)(cb * kCB)) * sizeof(float);
            const __amdgpu_buffer_rsrc_t xb = pc_rsrc(reinterpret_cast<const char*>(a.x) + base);
            const __amdgpu_buffer_rsrc_t rb = pc_rsrc(reinterpret_cast<const char*>(RES ? a.res : a.x) + base);
#pragma unroll
            for (int u = 0; u < kPcNPFx; ++u) {
                const unsigned o = nx ? nxt_off[u] : cur_off[u];
                r.pre[u] = pc_bload(xb, o);
                if constexpr (RES) r.prer[u] = pc_bload(rb, o);
            }
        };
        setup(tl, cur_off, cur_keep, cur_own);
        Regs set0, set1;
        issue(false, 0, set0);
        issue(false, 1, set1);               // NS >= 2 (checked by the launcher)
        int qbuf = 0;
#ifdef NRGBD_DEV
        long t_pub = 0, t_tr = 0, t_pbar = 0, t_q0 = 0, t_q1 = 0, t_q2 = 0;
#endif
        bool has_next = false;
        bool interior = false;   // the current tile's whole halo lies inside the image (set per tile below)
        int gi = 0;              // iterations so far (SH: strip parity; the transform lags one iteration)
        // (3) input transform B^T d B of this lane's (tile, word): rows (2 of the 4 xi_y), then columns; strip rawT -> V[qbuf]
        auto transform = [&]() __attribute__((always_inline)) {
            f32x4 ya[4], yb[4];
#pragma unroll
            for (int cc = 0; cc < 4; ++cc) {
                const int co = ((cc & 1) * 9 + (cc >> 1)) * kCB;
                const f32x4 R0 = *reinterpret_cast<const f32x4*>(rawT + rdR0 + co);
                const f32x4 R1 = *reinterpret_cast<const f32x4*>(rawT + rdR1 + co);
                const f32x4 R2 = *reinterpret_cast<const f32x4*>(rawT + rdR2 + co);
                ya[cc] = pk_fma_s(R1, m1, R0);   // R0 - R1
                yb[cc] = pk_fma_s(R2, sg, R1);     // R1 +- R2
            }
            float* Vq = Vb + qbuf * kPcV;
            const int xa = (2 * thalf) * 4, xb = (2 * thalf + 1) * 4;
            *reinterpret_cast<f32x4*>(Vq + pc_slot(xa + 0, ttile, tword)) = pk_fma_s(ya[2], m1, ya[0]);   // y0 - y2
            *reinterpret_cast<f32x4*>(Vq + pc_slot(xa + 1, ttile, tword)) = pk_add(ya[1], ya[2]);
            *reinterpret_cast<f32x4*>(Vq + pc_slot(xa + 2, ttile, tword)) = pk_fma_s(ya[1], m1, ya[2]);   // y2 - y1
            *reinterpret_cast<f32x4*>(Vq + pc_slot(xa + 3, ttile, tword)) = pk_fma_s(ya[3], m1, ya[1]);   // y1 - y3
            *reinterpret_cast<f32x4*>(Vq + pc_slot(xb + 0, ttile, tword)) = pk_fma_s(yb[2], m1, yb[0]);
            *reinterpret_cast<f32x4*>(Vq + pc_slot(xb + 1, ttile, tword)) = pk_add(yb[1], yb[2]);
            *reinterpret_cast<f32x4*>(Vq + pc_slot(xb + 2, ttile, tword)) = pk_fma_s(yb[1], m1, yb[2]);
            *reinterpret_cast<f32x4*>(Vq + pc_slot(xb + 3, ttile, tword)) = pk_fma_s(yb[3], m1, yb[1]);
        };
        // one stage: publish set r (stage s of the current tile), refill it for stage s+2, transform, barrier
        auto stage = [&](int s, Regs& r) __attribute__((always_inline)) {
#ifdef NRGBD_DEV
            const long p0 = wall_clock64();
            long p1 = p0;
#endif
            const int cb = s / KD, kd = s - cb * KD;
            const int z = KD == 3 ? tl.n + kd - 1 : tl.n;
            const bool zin = KD != 3 || (z >= 0 && z < a.N);
            if constexpr (SH) { raw = rawb + (gi & 1) * kPcShStrip; rawT = rawb + ((gi & 1) ^ 1) * kPcShStrip; }
            if (!(abl & 2)) {
#ifdef NRGBD_DEV
                if (abl & 64) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
                const long q0 = wall_clock64();
#endif
                if (!(abl & 8)) {   // (1) normalise / activate the prefetched words and publish them to this wave's strip
                    // Straight-line, packed, no per-lane branches or selects: as a chain of exec-masked blocks this phase took
                    // 0.74 us alone and 2.0 us beside the consumers' MFMA stream — longer than the MFMAs it has to stay ahead of.
                    if (!zin) {   // a depth tap outside the volume: the whole slice is zero padding
#pragma unroll
                        for (int u = 0; u < kPcNPFx; ++u) *reinterpret_cast<f32x4*>(raw + wr_off[u]) = f32x4{0.f, 0.f, 0.f, 0.f};
                    } else {
                        // (scale, shift) pairs re-paired for the packed FMAs: channels (0,1) and (2,3); identity = (1, 0)
                        asm volatile("" : "+v"(r.ss[0]), "+v"(r.ss[1]));   // re-paired HERE, not behind their loads
                        if constexpr (RES) asm volatile("" : "+v"(r.rs[0]), "+v"(r.rs[1]));
                        const f32x2 sc01 = r.ss[0].lo, sh01 = r.ss[0].hi, sc23 = r.ss[1].lo, sh23 = r.ss[1].hi;   // the LDS table is stored pre-paired (pc_ss_slot)
                        const f32x2 rc01 = r.rs[0].lo, rh01 = r.rs[0].hi, rc23 = r.rs[1].lo, rh23 = r.rs[1].hi;
                        const bool wmat = MAT && (KD != 3 || kd == 1) && tl.cg == 0;
                        // Breadth-first over the 5 items — all FMAs, then all ReLUs, then all masks, then the stores — and
                        // pinned in that order: beside the consumer's MFMA stream a VALU instruction that has to wait for
                        // its predecessor's result loses the issue port to the next MFMA (32 cycles), an independent one
                        // issues back to back.
                        // (with a residual operand: in two groups of items, which keeps the phase inside the register budget)
                        auto group = [&](auto u0_tag, auto u1_tag, auto interior_tag) __attribute__((always_inline)) {
                            constexpr int U0 = decltype(u0_tag)::value, U1 = decltype(u1_tag)::value, NU = U1 - U0;
                            constexpr bool INTERIOR = decltype(interior_tag)::value;   // every item of every lane inside the image: no padding mask
                            f32x2 lo[NU], hi[NU];
                            if constexpr (EPI == 1) {   // the R-Net form has no prologue: no identity FMAs (x * 1 + 0 is not folded: -0)
#pragma unroll
                                for (int i = 0; i < NU; ++i) { lo[i] = r.pre[U0 + i].lo; hi[i] = r.pre[U0 + i].hi; }
                            } else {
#pragma unroll
                                for (int i = 0; i < NU; ++i) {
                                    lo[i] = __builtin_elementwise_fma(r.pre[U0 + i].lo, sc01, sh01);
                                    hi[i] = __builtin_elementwise_fma(r.pre[U0 + i].hi, sc23, sh23);
                                }
                            }
                            __builtin_amdgcn_sched_barrier(0);
                            if (EPI == 0 && a.x_relu) {
#pragma unroll
                                for (int i = 0; i < NU; ++i) { lo[i].x = relu1(lo[i].x); lo[i].y = relu1(lo[i].y); hi[i].x = relu1(hi[i].x); hi[i].y = relu1(hi[i].y); }
                            }
                            __builtin_amdgcn_sched_barrier(0);
                            if constexpr (RES) {
                                f32x2 ql[NU], qh[NU];
#pragma unroll
                                for (int i = 0; i < NU; ++i) {
                                    ql[i] = __builtin_elementwise_fma(r.prer[U0 + i].lo, rc01, rh01);
                                    qh[i] = __builtin_elementwise_fma(r.prer[U0 + i].hi, rc23, rh23);
                                }
                                __builtin_amdgcn_sched_barrier(0);
                                if (a.res_relu) {
#pragma unroll
                                    for (int i = 0; i < NU; ++i) { ql[i].x = relu1(ql[i].x); ql[i].y = relu1(ql[i].y); qh[i].x = relu1(qh[i].x); qh[i].y = relu1(qh[i].y); }
                                }
                                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                                for (int i = 0; i < NU; ++i) { lo[i] = lo[i] + ql[i]; hi[i] = hi[i] + qh[i]; }
                                __builtin_amdgcn_sched_barrier(0);
                            }
                            // zero padding applies to the ACTIVATED tensor: out-of-image lanes (their loads read a harmless
                            // in-tensor word) are multiplied by 0
                            if constexpr (!INTERIOR) {
#pragma unroll
                                for (int i = 0; i < NU; ++i) {
                                    const f32x2 kk = {cur_keep[U0 + i], cur_keep[U0 + i]};
                                    lo[i] = lo[i] * kk; hi[i] = hi[i] * kk;
                                }
                                __builtin_amdgcn_sched_barrier(0);
                            }
#pragma unroll
                            for (int i = 0; i < NU; ++i) {
                                const int u = U0 + i;
                                const f32x4 v = __builtin_shufflevector(lo[i], hi[i], 0, 1, 2, 3);
                                // the activated input is written once: by the wave that owns the pixel, at the centre tap
                                if (MAT && wmat) {
                                    if ((cur_own >> u) & 1u)
                                        *reinterpret_cast<f32x4*>(reinterpret_cast<char*>(a.mat) + ((size_t)(KD == 3 ? z : 0) * plane + (size_t)(cb * kCB)) * sizeof(float) + cur_off[u]) = v;
                                }
                                *reinterpret_cast<f32x4*>(raw + wr_off[u]) = v;
                            }
                        };
                        // a tile whose halo lies inside the image needs no zero-padding mask (its pad pixels of the strip — the
                        // fifth item of lanes 32..63 — are never read by the transform)
                        if (interior) {
                            if constexpr (RES && kPcNPFx > 3) {
                                group(std::integral_constant<int, 0>{}, std::integral_constant<int, 3>{}, std::true_type{});
                                group(std::integral_constant<int, 3>{}, std::integral_constant<int, kPcNPFx>{}, std::true_type{});
                            } else {
                                group(std::integral_constant<int, 0>{}, std::integral_constant<int, kPcNPFx>{}, std::true_type{});
                            }
                        } else if constexpr (RES && kPcNPFx > 3) {
                            group(std::integral_constant<int, 0>{}, std::integral_constant<int, 3>{}, std::false_type{});
                            group(std::integral_constant<int, 3>{}, std::integral_constant<int, kPcNPFx>{}, std::false_type{});
                        } else {
                            group(std::integral_constant<int, 0>{}, std::integral_constant<int, kPcNPFx>{}, std::false_type{});
                        }
                    }
                }
#ifdef NRGBD_DEV
                const long q1 = wall_clock64();
#endif
                // (2) refill the set: stage s+2 of this tile, or stage s+2-NS of the next one
                {
                    // UNCONDITIONAL (the last two stages of the last tile re-read this tile's first two: harmless, never used)
                    const bool nx = s + 2 >= NS;
                    issue(nx && has_next, nx ? s + 2 - NS : s + 2, r);
                }
#ifdef NRGBD_DEV
                const long q2 = wall_clock64();
#endif
                if constexpr (!SH) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the strip is wave-private: in-order LDS, no barrier
#ifdef NRGBD_DEV
                p1 = wall_clock64();
                t_q0 += q0 - p0; t_q1 += q1 - q0; t_q2 += q2 - q1;
#endif
                if (!(abl & 4) && (!SH || gi > 0)) transform();   // SH: of the stage published one iteration ago
            }
#ifdef NRGBD_DEV
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            const long p2 = wall_clock64();
#endif
            __syncthreads();
#ifdef NRGBD_DEV
            t_pub += p1 - p0; t_tr += p2 - p1; t_pbar += wall_clock64() - p2;
#endif
            if (!SH || gi > 0) qbuf = qbuf == kPcNBuf - 1 ? 0 : qbuf + 1;
            ++gi;
        };
        for (int it = 0; it < count; ++it) {
            has_next = it + 1 < count;
            interior = tl.y0 + tl.py - DIL >= 0 && tl.y0 + tl.py + DIL * kPcTH < a.H && tl.x0 + tl.px - DIL >= 0 && tl.x0 + tl.px + DIL * kPcTW < a.W;
            if constexpr (!ODD) {   // stages in pairs: the register set of a stage is static
                for (int s = 0; s < NS; s += 2) {
                    // the book of the next tile is needed from the first refill that reaches into it (stage NS-2 refills stage 0)
                    if (s + 2 == NS && has_next) { tn = pc_decode<KD, DIL>(first + (it + 1) * step, a); setup(tn, nxt_off, nxt_keep, nxt_own); }
                    stage(s, set0);
                    stage(s + 1, set1);
                }
            } else {                // odd stage count (16 input channels x 3 depth taps): set = parity of the running stage count
                for (int s = 0; s < NS; ++s) {
                    if (s + 2 == NS && has_next) { tn = pc_decode<KD, DIL>(first + (it + 1) * step, a); setup(tn, nxt_off, nxt_keep, nxt_own); }
                    if (((unsigned)it * (unsigned)NS + (unsigned)s) & 1u) stage(s, set1);
                    else stage(s, set0);
                }
            }
            tl = tn;
#pragma unroll
            for (int u = 0; u < kPcNPFx; ++u) { cur_off[u] = nxt_off[u]; cur_keep[u] = nxt_keep[u]; }
            cur_own = nxt_own;
        }
        if constexpr (SH) {                    // the last published stage
            rawT = rawb + ((gi & 1) ^ 1) * kPcShStrip;
            if (!(abl & (2 | 4))) transform();
            __syncthreads();
        }
        __syncthreads();                       // the consumers' last two stages
        __syncthreads();
#ifdef NRGBD_DEV
        if ((abl & 64) && a.stats && wv == 0 && lane == 0) {
            float* o = a.stats + (size_t)blockIdx.x * 8 + 4;
            o[0] = (float)t_pub; o[1] = (float)t_tr; o[2] = (float)t_pbar; o[3] = 0.f;
            float* o2 = a.stats + 4096 + (size_t)blockIdx.x * 4;
            o2[0] = (float)t_q0; o2[1] = (float)t_q1; o2[2] = (float)t_q2; o2[3] = 0.f;
        }
#endif
    }
}

// Column-major partials [2C][rows] -> BatchNorm (scale, shift) [C][2] + running statistics: workgroup c reads channel c's two
// runs of `rows` floats coalesced (the row-major finaliser walks a 4-byte column of a [rows][2C] matrix: 64 us per K-Net layer
// at 24,576 rows; this one 6 us), fixed-order fp64 tree.
__global__ __launch_bounds__(1024) void bn_finalize_cm_kernel(const float* __restrict__ stats, int rows, int C, double count,
                                                              const float* __restrict__ gamma, const float* __restrict__ beta,
                                                              float eps, float momentum, float* __restrict__ running_mean,
                                                              float* __restrict__ running_var, float* __restrict__ ss, unsigned int* __restrict__ collapse_count, long long* __restrict__ batches_tracked) {
    __shared__ double sh[2][1024];
    const int c = blockIdx.x, tid = threadIdx.x;
    const float* p1 = stats + (size_t)c * rows;
    const float* p2 = stats + (size_t)(C + c) * rows;
    // a thread's share is a few dozen strided floats (K-Net at config B: rows = 12,288 -> 12 per run): all of them are requested
    // before the first is added (round 6: the one-at-a-time loop cost a DRAM latency per element, 15.4 us per K-Net layer at
    // config B); the order of the fp64 additions per thread is unchanged (g ascending), so are the bits
    double s1 = 0.0, s2 = 0.0;
    int g = tid;
    for (; g + 7 * 1024 < rows; g += 8 * 1024) {
        float u[8], v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) { u[k] = p1[g + k * 1024]; v[k] = p2[g + k * 1024]; }
#pragma unroll
        for (int k = 0; k < 8; ++k) { s1 += (double)u[k]; s2 += (double)v[k]; }
    }
    for (; g < rows; g += 1024) { s1 += (double)p1[g]; s2 += (double)p2[g]; }
    sh[0][tid] = s1; sh[1][tid] = s2;
    __syncthreads();
    for (int o = 512; o > 0; o >>= 1) {
        if (tid < o) { sh[0][tid] += sh[0][tid + o]; sh[1][tid] += sh[1][tid + o]; }
        __syncthreads();
    }
    if (tid == 0) bn_finalize_channel(sh[0][0], sh[1][0], count, gamma[c], beta[c], eps, momentum, running_mean, running_var, ss, c, collapse_count);
    if (tid == 0 && c == 0 && batches_tracked) *batches_tracked += 1;      // nn.BatchNorm's num_batches_tracked side effect (one launch less per layer)
}

// w [Cout][Cin][KD][3][3] -> U = G g G^T (float64, rounded once) in the kernel's B-operand order
// [cg][stage = cb*KD + kd][xi][wave][lane = kq*16 + j][e], co = cg*64 + 16*wave + j, ci = cb*16 + 4*kq + e
// transposed == 2 (grid.y = 2): BOTH streams of a layer [Cout][Cin] in one launch — y = 0 the forward one at wp, y = 1 the data-gradient
// one (roles of Cin / Cout swapped, taps flipped) right behind it (training packs both every iteration: the weights just changed)
__global__ __launch_bounds__(256) void conv_wino_pack_kernel(const float* __restrict__ w, float* __restrict__ wp, int Cin, int Cout, int KD,
                                                             int transposed) {
    const long total = (long)Cout * Cin * KD * 16;
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
    const int kd = (int)(t % KD); t /= KD;
    const int ncb = Cin / kCB;
    const int cb = (int)(t % ncb);
    const int cg = (int)(t / ncb);
    const int co = cg * 64 + 16 * wave + j, ci = cb * kCB + 4 * kq + e;
    // transposed: the stored tensor is [Cin][Cout][KD][3][3] (this kernel's ci is ITS output channel), taps flipped in every dimension
    const float* g = transposed ? w + (((size_t)ci * Cout + co) * KD + (KD - 1 - kd)) * 9 : w + (((size_t)co * Cin + ci) * KD + kd) * 9;
    const double G[4][3] = {{1.0, 0.0, 0.0}, {0.5, 0.5, 0.5}, {0.5, -0.5, 0.5}, {0.0, 0.0, 1.0}};
    const int a = xi >> 2, b = xi & 3;
    double u = 0.0;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) u += G[a][ky] * (double)g[transposed ? (2 - ky) * 3 + (2 - kx) : ky * 3 + kx] * G[b][kx];
    wp[idx] = (float)u;
}

}  // namespace nrgbd

extern "C" int nrgbd_conv_wino_pack(const float* w, float* w_wino, int Cin, int Cout, int kd, int transposed, void* stream) {
    using namespace nrgbd;
    if (!w || !w_wino) return NRGBD_E_NULL;
    if (Cin <= 0 || Cin % kCB || Cout <= 0 || Cout % 64 || (kd != 1 && kd != 3)) return NRGBD_E_SHAPE;
    if (transposed < 0 || transposed > 2) return NRGBD_E_ARG;
    if (transposed == 2 && Cin % 64) return NRGBD_E_SHAPE;           // both streams: each channel count is a Cout once
    const long total = (long)Cout * Cin * kd * 16;
    hipLaunchKernelGGL(conv_wino_pack_kernel, dim3((unsigned)((total + 255) / 256), transposed == 2 ? 2 : 1), dim3(256), 0, (hipStream_t)stream,
                       w, w_wino, Cin, Cout, kd, transposed);
    NRGBD_CHECK_LAUNCH();
    return NRGBD_OK;
}

extern "C" int nrgbd_bn_finalize_cm(const float* stats, int rows, int C, long count, const float* gamma, const float* beta,
                                    float eps, float momentum, float* running_mean, float* running_var, float* scale_shift, unsigned int* collapse_count, long long* batches_tracked,
                                    void* stream) {
    using namespace nrgbd;
    if (!stats || !gamma || !beta || !scale_shift) return NRGBD_E_NULL;
    if (rows <= 0 || count <= 0 || C <= 0) return NRGBD_E_SHAPE;
    if ((running_mean == nullptr) != (running_var == nullptr)) return NRGBD_E_NULL;
    hipLaunchKernelGGL(bn_finalize_cm_kernel, dim3(C), dim3(1024), 0, (hipStream_t)stream, stats, rows, C, (double)count, gamma,
                       beta, eps, momentum, running_mean, running_var, scale_shift, collapse_count, batches_tracked);
    NRGBD_CHECK_LAUNCH();
    return NRGBD_OK;
}

extern "C" int nrgbd_conv_wino_tiles(int N, int H, int W, int dilation) {
    using namespace nrgbd;
    if (N <= 0 || H <= 0 || W <= 0 || (dilation != 1 && dilation != 2)) return NRGBD_E_SHAPE;
    return N * ceil_div(H, kPcTH * dilation) * ceil_div(W, kPcTW * dilation) * dilation * dilation;
}

extern "C" int nrgbd_conv_wino_f32(const float* x, const float* x_ss, int x_relu, const float* res, const float* res_ss,
                                   int res_relu, float* materialized, const float* w_wino, float* y, float* stats, int N,
                                   int H, int W, int Cin, int Cout, int kd, int dilation, void* stream) {
    using namespace nrgbd;
    if (!x || !w_wino || !y) return NRGBD_E_NULL;
    const bool half = Cout == 32;               // the HALF form: 2-D, dilation 1 only
    if (N <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cin % kCB || Cin > 2048 || Cout <= 0 || (Cout % 64 && !half)) return NRGBD_E_SHAPE;
    if ((kd != 1 && kd != 3) || (dilation != 1 && dilation != 2) || (kd == 3 && dilation != 1)) return NRGBD_E_ARG;
    if (half && (kd != 1 || dilation != 1)) return NRGBD_E_SHAPE;
    if ((Cin / kCB) * kd < 2) return NRGBD_E_SHAPE;      // two stages are always in flight (two register sets)
    // 32-bit BYTE offsets in the loader: inside one slice when kd = 3 (the slice is a 64-bit base), inside the tensor otherwise
    if ((long)(kd == 3 ? 1 : N) * H * W * Cin >= (1L << 30)) return NRGBD_E_SHAPE;
    const int rows = nrgbd_conv_wino_tiles(N, H, W, dilation);
    const long nt = (long)rows * (half ? 1 : Cout / 64);
    if (nt >= (1L << 30)) return NRGBD_E_SHAPE;
    WinoPcArgs a{x, x_ss, res, res_ss, materialized, w_wino, y, stats, x_relu, res_relu, N, H, W, Cin, Cout, (int)nt, half ? 2 * rows : rows,
                 nullptr, 0, 0, 0, 0, dev_env_int("NRGBD_WINO_ABL")};
    int dev = 0, ncu = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e == hipSuccess) e = hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev);
    if (e != hipSuccess) return (int)e;
    if (ncu <= 0) return NRGBD_E_ARG;
    const int nwg = nt < ncu ? (int)nt : ncu;   // persistent: one workgroup per CU
    const size_t lds = (size_t)(kPcNBuf * kPcV + kPcStrips + 4 * Cin) * sizeof(float);   // 96 KB V + 25.6 (20) KB strips + (scale, shift) tables
    // The opt-in for > 64 KB of dynamic LDS is a property of the FUNCTION, and a launch recorded in a hipGraph is replayed under whatever
    // value the function carries at that moment: it is therefore always set to the form's maximum (Cin = 2048), never to this call's size —
    // a later eager call with fewer channels would otherwise shrink it under a captured launch with more (round 6: a training graph
    // replayed after an eager iteration ran with part of its (scale, shift) tables cut off)
    const int lds_attr = (int)((size_t)(kPcNBuf * kPcV + kPcStrips + 4 * 2048) * sizeof(float));
    hipStream_t st = (hipStream_t)stream;
#define NRGBD_WINO_PC_LAUNCH(KD_, DIL_, RES_)                                                                       \
    do { if (materialized) NRGBD_WINO_PC_LAUNCH_M(KD_, DIL_, RES_, true); else NRGBD_WINO_PC_LAUNCH_M(KD_, DIL_, RES_, false); } while (0)
#define NRGBD_WINO_PC_LAUNCH_M(KD_, DIL_, RES_, MAT_)                                                               \
    do {                                                                                                            \
        /* > 64 KB of dynamic LDS needs the opt-in: once per function and device (common.hpp set_max_dynamic_lds)       */ \
        e = set_max_dynamic_lds(reinterpret_cast<const void*>(&conv_wino_pc_kernel<KD_, DIL_, RES_, false, 0, MAT_>), \
                                lds_attr);                              \
        if (e != hipSuccess) return (int)e;                                                                         \
        hipLaunchKernelGGL((conv_wino_pc_kernel<KD_, DIL_, RES_, false, 0, MAT_>), dim3(nwg), dim3(512), lds, st, a); \
    } while (0)
    const bool odd = (((Cin / kCB) * kd) & 1) != 0;
    if (odd && (kd != 3 || res || materialized)) return NRGBD_E_SHAPE;   // an odd stage count is instantiated for the K-Net's first layer only
#define NRGBD_WINO_PC_LAUNCH_H(RES_, MAT_)                                                                          \
    do {                                                                                                            \
        e = set_max_dynamic_lds(reinterpret_cast<const void*>(&conv_wino_pc_kernel<1, 1, RES_, false, 0, MAT_, true>), \
                                lds_attr);                              \
        if (e != hipSuccess) return (int)e;                                                                         \
        hipLaunchKernelGGL((conv_wino_pc_kernel<1, 1, RES_, false, 0, MAT_, true>), dim3(nwg), dim3(512), lds, st, a); \
    } while (0)
    if (half) {
        if (res && materialized) NRGBD_WINO_PC_LAUNCH_H(true, true);
        else if (res) NRGBD_WINO_PC_LAUNCH_H(true, false);
        else if (materialized) NRGBD_WINO_PC_LAUNCH_H(false, true);
        else NRGBD_WINO_PC_LAUNCH_H(false, false);
    } else if (kd == 3 && odd) {
        e = set_max_dynamic_lds(reinterpret_cast<const void*>(&conv_wino_pc_kernel<3, 1, false, true>), lds_attr);
        if (e != hipSuccess) return (int)e;
        hipLaunchKernelGGL((conv_wino_pc_kernel<3, 1, false, true>), dim3(nwg), dim3(512), lds, st, a);
    } else if (kd == 3) {
        if (res) NRGBD_WINO_PC_LAUNCH(3, 1, true); else NRGBD_WINO_PC_LAUNCH(3, 1, false);
    } else if (dilation == 1) {
        if (res) NRGBD_WINO_PC_LAUNCH(1, 1, true); else NRGBD_WINO_PC_LAUNCH(1, 1, false);
    } else {
        if (res) NRGBD_WINO_PC_LAUNCH(1, 2, true); else NRGBD_WINO_PC_LAUNCH(1, 2, false);
    }
#undef NRGBD_WINO_PC_LAUNCH
#undef NRGBD_WINO_PC_LAUNCH_M
#undef NRGBD_WINO_PC_LAUNCH_H
    NRGBD_CHECK_LAUNCH();
    return NRGBD_OK;
}

// R-Net form (models/m_submodule.py:18-27 conv2d_leakyRelu at widths with Cin % 16 == 0 (>= 32), Cout % 64 == 0: Refine.py:51-56 conv0,
// conv0_1): 3x3 convolution + bias + LeakyReLU(0.01) in the Winograd domain, no prologue, no statistics
extern "C" int nrgbd_conv_wino_rnet_ex_f32(const float* x, const float* w_wino, const float* bias, int out_lrelu, float* y, int N,
                                           int H, int W, int Cin, int Cout, int ldy, int ycoff, int cout_valid, void* stream) {
    using namespace nrgbd;
    if (!x || !w_wino || !y) return NRGBD_E_NULL;
    const bool half = Cout == 32;               // the HALF form: a 32-column slice (e.g. the 3 columns a 67-wide layer has beyond 64)
    if (N <= 0 || H <= 0 || W <= 0 || Cin < 32 || Cin % kCB || Cout <= 0 || (Cout % 64 && !half)) return NRGBD_E_SHAPE;   // >= 2 stages
    if ((long)N * H * W * Cin >= (1L << 30)) return NRGBD_E_SHAPE;
    if (cout_valid < 0 || cout_valid > Cout || ycoff < 0 || (ldy != 0 && ldy < ycoff + (cout_valid ? cout_valid : Cout))) return NRGBD_E_ARG;
    if ((long)N * H * W * (ldy ? ldy : Cout) >= (1L << 32)) return NRGBD_E_SHAPE;   // 32-bit lane offsets into a tile's rows only, but keep it sane
    const int rows = nrgbd_conv_wino_tiles(N, H, W, 1);
    const long nt = (long)rows * (half ? 1 : Cout / 64);
    if (nt >= (1L << 30)) return NRGBD_E_SHAPE;
    WinoPcArgs a{x, nullptr, nullptr, nullptr, nullptr, w_wino, y, nullptr, 0, 0, N, H, W, Cin, Cout, (int)nt, half ? 2 * rows : rows,
                 bias, out_lrelu, ldy, ycoff, cout_valid, 0};
    int dev = 0, ncu = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e == hipSuccess) e = hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev);
    if (e != hipSuccess) return (int)e;
    if (ncu <= 0) return NRGBD_E_ARG;
    const int nwg = nt < ncu ? (int)nt : ncu;
    const size_t lds = (size_t)(kPcNBuf * kPcV + kPcStrips) * sizeof(float);
#define NRGBD_WINO_RNET_LAUNCH(ODD_, HALF_)                                                                              \
    do {                                                                                                                 \
        e = set_max_dynamic_lds(reinterpret_cast<const void*>(&conv_wino_pc_kernel<1, 1, false, ODD_, 1, false, HALF_>), \
                                (int)lds);                                   \
        if (e != hipSuccess) return (int)e;                                                                              \
        hipLaunchKernelGGL((conv_wino_pc_kernel<1, 1, false, ODD_, 1, false, HALF_>), dim3(nwg), dim3(512), lds, (hipStream_t)stream, a); \
    } while (0)
    // an odd stage count (the R-Net's 67 -> 80 and 131 -> 144 channel pixels): the register set of a stage = parity of the running count
    const bool odd = ((Cin / kCB) & 1) != 0;
    if (odd && half) NRGBD_WINO_RNET_LAUNCH(true, true);
    else if (odd) NRGBD_WINO_RNET_LAUNCH(true, false);
    else if (half) NRGBD_WINO_RNET_LAUNCH(false, true);
    else NRGBD_WINO_RNET_LAUNCH(false, false);
#undef NRGBD_WINO_RNET_LAUNCH
    NRGBD_CHECK_LAUNCH();
    return NRGBD_OK;
}

extern "C" int nrgbd_conv_wino_rnet_f32(const float* x, const float* w_wino, const float* bias, int out_lrelu, float* y, int N,
                                        int H, int W, int Cin, int Cout, void* stream) {
    return nrgbd_conv_wino_rnet_ex_f32(x, w_wino, bias, out_lrelu, y, N, H, W, Cin, Cout, 0, 0, 0, stream);
}
