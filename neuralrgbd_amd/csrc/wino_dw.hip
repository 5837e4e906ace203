// wino_dw.hip — the K-Net's 3x3x3 convolutions (models/basic.py:71-94) in the Winograd domain in ALL THREE dimensions:
// F(2x2, 3x3) in the image plane (as wino_pc.hip) and F(2, 3) along the depth axis on top of it.
//
// Why.  wino_pc.hip treats the three depth taps of a 3x3x3 layer as three independent 2-D Winograd problems: 16 multiplies per
// 2x2 outputs and depth tap, 3 taps -> 12 multiplies per output voxel (direct: 27).  At config B its ten 64 -> 64 layers are
// 61 % of the frame and run at 66 % of the fp32 matrix peak with the matrix pipe as the critical resource — the only lever left
// is fewer multiplies.  F(2, 3) along depth produces TWO output slices from FOUR transformed slices: 4 x 16 multiplies per
// 2x2x2 outputs = 8 per output voxel, 1.5x fewer MFMAs, still exact-algorithm fp32 (only the rounding order differs; measured
// against float64: mean error 1.3x the 2-D form's, tests/test_gpu_knet.py).
//
//   input   d_j = act(x[z0 - 1 + j]),  j = 0..3                      (z0 = first of the two output slices of a tile)
//   depth   D_0 = d_0 - d_2,  D_1 = d_1 + d_2,  D_2 = d_2 - d_1,  D_3 = d_1 - d_3          (B^T of F(2,3))
//   plane   V_t = B^T D_t B per 4x4 patch                                                   (as wino_pc.hip)
//   weights U_t = sum_kd Gd[t][kd] (G g_kd G^T),  Gd = [1 0 0; .5 .5 .5; .5 -.5 .5; 0 0 1]  (packed on the device, float64)
//   product M_t = sum_ci V_t U_t                                                            (the MFMAs: 4 x Cin/16 stages)
//   output  y[z0] = A^T (M_0 + M_1 + M_2) A,   y[z0 + 1] = A^T (M_1 - M_2 - M_3) A
//
// Same persistent producer / consumer organisation, tile geometry, LDS images and weight-stream layout as wino_pc.hip (read its
// header first); what differs:
//   * a tile is 8x16 pixels x TWO depth slices; its stage list is t-major: stage s = t * (Cin/16) + cb, so that the consumers'
//     128 accumulator registers hold ONE M_t at a time;
//   * consumers: at the end of phase t (its last channel block) the accumulators are inverse-transformed in the plane (A^T . A:
//     32 values per lane) and folded into the two output slices, which accumulate in two 32 KB LDS stashes (the register file
//     is full: 128 accumulators + weight ring + operands); phase 2 completes and stores slice z0, phase 3 slice z0+1, each
//     with its BatchNorm partial statistics;
//   * producers: a stage needs the depth combination of TWO slices.  The unit of prefetch stays one (slice, channel block):
//     unit A is normalised / activated and published to the stage's strip exactly as in wino_pc.hip, unit B is activated and
//     COMBINED with what the strip holds (A +- B: a lane reads back its own words; a wave's LDS operations execute in order),
//     and the strip is transformed one iteration later (shared strips, below).  Each unit has its own register set, refilled
//     for the NEXT stage right after it was published.
// Work per pair of output slices: 16 stages (wino_pc: 24); producer publishes 32 (24); plane transforms 16 (24).
// LDS: 2 x 32 KB V + 2 x 12.8 KB strips + 2 x 32 KB stash = 154 KB; 8 waves, up to 256 VGPRs each.  (Two V buffers instead of
// wino_pc's three — the third one's 32 KB hold the second stash.  A consumer therefore cannot read the first operand of the
// next stage before the stage barrier; instead it ARRIVES at the barrier early, as soon as its last LDS operand of the stage is
// in registers, and reads the next stage's first operand under its own last 8 MFMAs.)
#include <type_traits>

#include "wino_pc.hpp"

namespace nrgbd {

constexpr int kDwStashWave = 2 * 8 * 64 * 4;   // floats of one consumer wave's stash: [2 output slices][8 words][64 lanes][4]
constexpr int kDwMaxCin = 512;                 // (scale, shift) tables of x and res live in LDS: 2 x 2 x Cin floats
constexpr int kDwNBuf = 2;                     // V buffers (wino_pc.hip: 3; the third one's 32 KB hold the second stash here)
// SHARED strips (round 4): wino_pc.hip's producer wave p loads the four halo rows 2p .. 2p+3 its tile row needs into a PRIVATE strip
// — 16 rows for a 10-row halo, i.e. every interior row is loaded, activated and published twice.  Here the 10 x 18 halo of a unit is
// split once over the 256 producer lanes (3 words per lane instead of 5) into a strip all four waves share, published one stage
// AHEAD: iteration i publishes stage i into strip[i & 1] and transforms stage i - 1 from strip[(i - 1) & 1] (complete since the
// stage barrier), so the only synchronisation is the barrier the stage has anyway (one more at the start).  Per stage and
// producer wave: 6 instead of 10 loads, 18 instead of 30 packed activation / combine FMAs, 9 instead of 15 strip accesses.
#ifndef NRGBD_DW_SHARED
#define NRGBD_DW_SHARED 1   // 0: the private-strip producers (experimental A/B builds only)
#endif
constexpr int kDwShRows = kPcTH + 2;                       // halo rows of a tile
constexpr int kDwShStrip = kDwShRows * kPcRawW * kCB;      // floats of one shared strip: [10 rows][20 pixels][16] = 12.8 KB
constexpr int kDwShItems = kDwShRows * 18 * 4;             // (row, column, 16-byte word) items of a unit: 720
constexpr int kDwNPF = NRGBD_DW_SHARED ? 3 : kPcNPF;       // items per producer lane and unit
constexpr int kDwStrips = NRGBD_DW_SHARED ? 2 * kDwShStrip : 4 * kPcRawWave;   // floats of the strip region

struct DwTile { int z0, y0, x0, cg, row0; };   // row0: statistics row of slice z0 (slice z0 + 1: row0 + 1)

__device__ __forceinline__ DwTile dw_decode(int t, const WinoPcArgs& a) {
    DwTile r;
    const int ncg = a.Cout >> 6;
    const int tiles_x = (a.W + kPcTW - 1) / kPcTW;
    const int row = t / ncg;
    r.cg = t - row * ncg;
    t = row;
    const int npair = a.N >> 1;
    const int zp = t % npair; t /= npair;       // depth fastest: list neighbours share three of their four input slices
    const int tx = t % tiles_x, ty = t / tiles_x;
    r.z0 = 2 * zp;
    r.y0 = ty * kPcTH; r.x0 = tx * kPcTW;
    r.row0 = (ty * tiles_x + tx) * a.N + r.z0;
    return r;
}

// Channel block of the i-th stage of phase t: odd phases sweep the blocks backwards (ncb-1 .. 0).  Phases 1 and 2 read the same
// two slices, phase 0 / 1 and 2 / 3 share one: with every phase sweeping forwards a unit's re-read came Cin/16 stages after its
// first read, and the 32 workgroups of an XCD stream 1.5 MB (3 MB with a residual operand) per stage through their 4 MB L2 beside
// the 1 MB weight stream — every re-read missed (profiles/r3_pmc_wino.txt: the residual variant fetched ALL its reads).  Turning
// round at the phase boundary puts the most recently read units first.
#ifndef NRGBD_DW_IDENT
#define NRGBD_DW_IDENT 1   // 0: experimental A/B builds only (tools/knet_ab.py)
#endif
#ifndef NRGBD_DW_SERP
#define NRGBD_DW_SERP 1
#endif
__device__ __forceinline__ int dw_cb(int t, int i, int ncb) { return (NRGBD_DW_SERP && (t & 1)) ? ncb - 1 - i : i; }

// slices combined by stage phase t: V_t = d[zA] + sign * d[zB]
__device__ __forceinline__ int dw_zA(int t) { return t == 0 ? -1 : (t == 2 ? 1 : 0); }   // relative to z0: -1, 0, 1, 0
__device__ __forceinline__ int dw_zB(int t) { return t == 2 ? 0 : (t == 3 ? 2 : 1); }    //                  1, 1, 0, 2

// RES: a second operand (res) is added after activation.  MAT: the activated input is also written out (a.mat).  MAT is a
// template parameter because of what its stores do to the OTHER variants: with loads and stores of one wave both pending the
// compiler cannot rely on in-order completion and turns every s_waitcnt on a prefetched register into vmcnt(0) — which also
// waits for the refill issued a few instructions earlier, i.e. exposes a full memory latency per stage.
// RSID: the residual operand comes with identity (scale, shift) and no ReLU (the K-Net's four residual layers: a materialised
// skip tensor) — its normalisation FMAs are dropped.
// IDENT: x needs no (scale, shift) and no ReLU (an input some earlier pass materialised: the K-Net's residual layers behind
// nrgbd_nhwc_act, every layer of the training path) — the producers' 10 packed FMAs per unit are dropped (instantiated for the
// plain form only).
// CLAMP: x = relu(x * s + t) with the ReLU taken by the FMA's own [0, 1] clamp: the (scale, shift) pairs are multiplied by
// a.x_unit = 2^-k on their way to LDS (k chosen by the caller so that no activated value can reach 2^k: |BatchNorm(y)| <=
// |gamma| sqrt(n) + |beta| for batch statistics over n values) and the weight stream carries 2^k.  Scaling by a power of two
// commutes with every rounding of the path, so the output bits are those of the plain form; the producers lose the 20
// v_max_f32 per unit (instantiated for the plain form only).
template <bool RES, bool MAT, bool RSID, bool IDENT = false, bool CLAMP = false>
__global__ __launch_bounds__(512) void conv_wino_dw_kernel(const WinoPcArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* Vb = lds;                                   // [2][16 xi][32 tiles][16]
    float* rawb = lds + kDwNBuf * kPcV;                // SHARED: [2][10 rows][20 pixels][16]; else [4 producer waves][4 rows][20 pixels][16]
    float* stashb = rawb + kDwStrips;                  // [4 consumer waves][2 slices][8][64][4]
    float* ssl = stashb + 4 * kDwStashWave;            // [Cin][2] (scale, shift) of x, then [Cin][2] of res

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wv = wave & 3;
    const int ncb = a.Cin / kCB;
    const int NS = 4 * ncb;                            // stages per tile (pair of slices)
#ifdef NRGBD_DEV
    const int abl = a.abl;   // developer ablations (results invalid): 1 no MFMAs, 2 no producer work, 4 no transform, 8 no publish A,
                             // 16 no publish B, 32 no refills, 64 no fold
#else
    constexpr int abl = 0;
#endif

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
    // The per-channel (scale, shift) pairs go to LDS once (identity where the pointer is null).  Loading a stage's pairs from
    // global memory with its raw words — as wino_pc.hip does — puts them FIRST in the refill's queue, and the compiler moves them
    // into their home registers immediately: an s_waitcnt right behind the loads, i.e. one exposed L2 round trip per unit.
    for (int i = tid; i < 2 * a.Cin; i += 512) {
        const int j = pc_ss_slot(i);      // pairs as the packed FMAs take them: (s0, s1, t0, t1 | s2, s3, t2, t3) per 4 channels
        ssl[j] = (a.x_ss ? a.x_ss[i] : ((i & 1) ? 0.f : 1.f)) * (CLAMP ? a.x_unit : 1.f);
        ssl[2 * a.Cin + j] = (RES && a.res_ss) ? a.res_ss[i] : ((i & 1) ? 0.f : 1.f);
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
        f32x4* stash0 = reinterpret_cast<f32x4*>(stashb + wv * kDwStashWave) + lane;     // slice z0:     word i at stash0[i * 64]
        f32x4* stash1 = stash0 + 8 * 64;                                                  // slice z0 + 1

        DwTile tl = dw_decode(first, a);
        const f32x4* wt = wbase + (size_t)tl.cg * wgroup;
        f32x4 Bn[kPcNB], An[2][2];
#pragma unroll
        for (int b = 0; b < kPcBD; ++b) Bn[b] = wt[b * 256];
        if constexpr (NRGBD_DW_SHARED != 0) __syncthreads();   // the producers publish stage 0 (transformed one iteration later)
        __syncthreads();                               // producers finish stage 0
        int buf = 0;
        An[0][0] = *reinterpret_cast<const f32x4*>(Vb + a0);
        An[0][1] = *reinterpret_cast<const f32x4*>(Vb + a1);
        float neg1 = -1.f;
        asm volatile("" : "+v"(neg1));
        const f32x2 n1 = {neg1, neg1};

        for (int it = 0; it < count; ++it) {
            const int tnext = first + (it + 1 < count ? it + 1 : it) * step;
            const DwTile tn = dw_decode(tnext, a);
            const f32x4* wt_next = wbase + (size_t)tn.cg * wgroup;
            const int co = tl.cg * 64 + wv * 16 + jj;
            // one phase = the Cin/16 stages of depth-transform index T (accumulating M_T), then its fold; the four phases are
            // separate straight-line instantiations so that the accumulators stay in fixed registers
            auto phase = [&](auto t_tag) __attribute__((always_inline)) {
                constexpr int T = decltype(t_tag)::value;
                for (int cb = 0; cb < ncb; ++cb) {
                    const int s = T * ncb + dw_cb(T, cb, ncb);          // cb: position in the phase's sweep
                    const float* Vc = Vb + buf * kPcV;
                    const int nbuf = buf ^ 1;
                    const float* Vn = Vb + nbuf * kPcV;
                    const f32x4* wcur = wt + (size_t)s * (16 * 256);
                    const f32x4* wnx = cb + 1 < ncb ? wt + (size_t)(T * ncb + dw_cb(T, cb + 1, ncb)) * (16 * 256)
                                       : (T < 3 ? wt + (size_t)((T + 1) * ncb + dw_cb(T + 1, 0, ncb)) * (16 * 256) : wt_next);
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
                                // the weight line of the point 7 ahead is requested HERE, in the second MFMA gap of the point, not at its top beside the two
                                // LDS reads: a vector-memory instruction costs the wave ~50 issue cycles, and three memory instructions in one gap let the
                                // matrix pipe run dry (tools/probes/mfma_stream_probe.hip: 78.5 -> 85.4 % busy)
                                if (e == NRGBD_WPOS) Bn[(xi + kPcBD) % kPcNB] = xi + kPcBD < 16 ? wcur[(xi + kPcBD) * 256] : wnx[(xi + kPcBD - 16) * 256];
                                __builtin_amdgcn_sched_barrier(0);
                            }
                            if (xi == 14) {
                                // EARLY stage barrier: the last LDS operands of this stage (xi = 15) are in registers, so the wave
                                // can let the producers overwrite this V buffer already — and read the first operand of the next
                                // stage (complete once the barrier is passed) under its own last 8 MFMAs.  With two V buffers this
                                // hides the LDS latency a third buffer hides in wino_pc.hip.
                                __syncthreads();
                                An[0][0] = *reinterpret_cast<const f32x4*>(Vn + a0);
                                An[0][1] = *reinterpret_cast<const f32x4*>(Vn + a1);
                                __builtin_amdgcn_sched_barrier(0);
                            }
                        }
                    };
                    if (!(abl & 1)) { if (cb == 0) body(std::true_type{}); else body(std::false_type{}); }
                    else {
                        __syncthreads();
                        An[0][0] = *reinterpret_cast<const f32x4*>(Vn + a0);
                        An[0][1] = *reinterpret_cast<const f32x4*>(Vn + a1);
                    }
                    buf = nbuf;
                }
                // ---- end of phase t: plane inverse transform of M_t (A^T . A) and the depth fold
                //   y[z0]     = M_0 + M_1 + M_2      (LDS stash 0; complete after phase 2)
                //   y[z0 + 1] = M_1 - M_2 - M_3      (LDS stash 1; complete after phase 3)
                // (in registers the 32 running values of a slice do not fit beside 128 accumulators + weight ring + operands: the
                //  compiler spilled them to scratch INSIDE the MFMA blocks, i.e. into the in-order queue of the weight loads)
                // lane (kq, jj): output channel co = 16 wv + jj; register r of row block m = Winograd tile 16 m + 4 kq + r = tile
                // row 2m + (kq >> 1), tile column 4 (kq & 1) + r; word (m, rp, aa) = output row 2 (tile row) + aa, tiles r = 2rp
                // (.x of a pair) and 2rp + 1 (.y), columns 2 (tile column) + {0: o0, 1: o1}
                if (!(abl & 64)) {
                    constexpr bool EMIT = T >= 2;
                    const int zs = tl.z0 + (T == 3 ? 1 : 0);
                    float* ybase = a.y + (((size_t)zs * a.H + tl.y0) * a.W + tl.x0) * a.Cout + tl.cg * 64 + wv * 16;
                    f32x2 S1 = {0.f, 0.f}, S2 = {0.f, 0.f};
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
                            for (int aa = 0; aa < 2; ++aa) {
                                const int wi = (m * 2 + rp) * 2 + aa;
                                f32x2 o0 = (tr[aa][0] + tr[aa][1]) + tr[aa][2];
                                f32x2 o1 = __builtin_elementwise_fma(tr[aa][3], n1, __builtin_elementwise_fma(tr[aa][2], n1, tr[aa][1]));
                                if constexpr (T == 0) {
                                    stash0[wi * 64] = __builtin_shufflevector(o0, o1, 0, 1, 2, 3);
                                } else if constexpr (T == 1) {
                                    const f32x4 o = __builtin_shufflevector(o0, o1, 0, 1, 2, 3);
                                    stash0[wi * 64] = stash0[wi * 64] + o;
                                    stash1[wi * 64] = o;
                                } else if constexpr (T == 2) {
                                    const f32x4 o = __builtin_shufflevector(o0, o1, 0, 1, 2, 3);
                                    const f32x4 b = stash1[wi * 64];
                                    stash1[wi * 64] = b - o;
                                    const f32x4 y = stash0[wi * 64] + o;
                                    o0 = y.lo; o1 = y.hi;
                                } else {
                                    const f32x4 o = __builtin_shufflevector(o0, o1, 0, 1, 2, 3);
                                    const f32x4 y = stash1[wi * 64] - o;
                                    o0 = y.lo; o1 = y.hi;
                                }
                                if constexpr (EMIT) {
                                    float* oa = ybase + ((size_t)(4 * m + aa) * a.W + (size_t)(2 * (2 * rp))) * a.Cout;       // tile r = 2 rp
                                    float* ob = ybase + ((size_t)(4 * m + aa) * a.W + (size_t)(2 * (2 * rp + 1))) * a.Cout;   // tile r + 1
                                    oa[lane_yoff] = o0.x; oa[lane_yoff + a.Cout] = o1.x;
                                    ob[lane_yoff] = o0.y; ob[lane_yoff + a.Cout] = o1.y;
                                    S1 = (S1 + o0) + o1;
                                    S2 = __builtin_elementwise_fma(o1, o1, __builtin_elementwise_fma(o0, o0, S2));
                                }
                            }
                            // one (m, rp) group at a time: left to itself the scheduler interleaves all four groups and the 8 stash
                            // words, and the register file (128 accumulators + ring + operands live here) overflows into scratch
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    }
                    if constexpr (EMIT) {
                        if (a.stats) {   // the wave owns its 16 channels: reduce over the 4 lanes (kq) that share a channel
                            float s1 = S1.x + S1.y, s2 = S2.x + S2.y;
                            s1 += __shfl_xor(s1, 16, 64); s2 += __shfl_xor(s2, 16, 64);
                            s1 += __shfl_xor(s1, 32, 64); s2 += __shfl_xor(s2, 32, 64);
                            if (kq == 0) {
                                const int row = tl.row0 + (T == 3 ? 1 : 0);
                                a.stats[(size_t)co * a.rows + row] = s1;
                                a.stats[(size_t)(a.Cout + co) * a.rows + row] = s2;
                            }
                        }
                    }
                }
            };
            phase(std::integral_constant<int, 0>{});
            phase(std::integral_constant<int, 1>{});
            phase(std::integral_constant<int, 2>{});
            phase(std::integral_constant<int, 3>{});
            tl = tn;
            wt = wt_next;
        }
    } else {
        // =========================================== producer: tile row pw (8 Winograd tiles) ===========================
        const int pw = wv;
        constexpr bool SH = NRGBD_DW_SHARED != 0;
        constexpr int kItems = SH ? kDwShItems : kPcItems;
        float* raw = SH ? rawb : rawb + pw * kPcRawWave;   // SH: the strip this iteration PUBLISHES into (set per iteration)
        const float* rawT = raw;                            // ... and the one it TRANSFORMS from
        const int w4 = lane & 3;
        // item u of this lane: word (item & 3) of halo pixel item >> 2, pixels in (row, de-interleaved column) order.
        // SH: item = 192 pw + lane + 64 u over the whole 10-row halo; else lane + 64 u over the wave's own 4 rows
        auto item_id = [&](int u) { return (SH ? 192 * pw : 0) + lane + 64 * u; };
        auto item_rr = [&](int u) { return (item_id(u) >> 2) / 18; };
        auto item_cp = [&](int u) { const int pi = item_id(u) >> 2; return pi - (pi / 18) * 18; };
        auto item_col = [&](int u) { const int cp = item_cp(u); return cp < 9 ? 2 * cp : 2 * cp - 17; };
        int wr_off[kDwNPF];
#pragma unroll
        for (int u = 0; u < kDwNPF; ++u) {
            const int item = item_id(u), e = (item - kItems) >> 2;   // lanes without an item write a zero into a pad pixel (columns 18, 19)
            wr_off[u] = item < kItems ? (item_rr(u) * kPcRawW + item_cp(u)) * kCB + w4 * 4
                                      : ((e >> 1) * kPcRawW + 18 + (e & 1)) * kCB + w4 * 4;
        }
        const int tword = lane & 3, txl = ((lane >> 5) << 2) | ((lane >> 2) & 3), thalf = (lane >> 4) & 1;
        const int ttile = pw * 8 + txl;
        const int rdc = txl * kCB + tword * 4 + (SH ? 2 * pw * kPcRawW * kCB : 0);   // SH: the tile row's halo rows start at strip row 2 pw
        const int rdR0 = (thalf ? 2 : 0) * kPcRawW * kCB + rdc, rdR1 = (thalf ? 1 : 2) * kPcRawW * kCB + rdc,
                  rdR2 = (thalf ? 3 : 1) * kPcRawW * kCB + rdc;
        const float sg = thalf ? -1.f : 1.f;
        float m1 = -1.f;
        asm volatile("" : "+v"(m1));

        unsigned cur_off[kDwNPF], cur_own = 0, nxt_off[kDwNPF], nxt_own = 0;   // BYTE offsets inside a slice
        float cur_keep[kDwNPF], nxt_keep[kDwNPF];
        auto setup = [&](const DwTile& tt, unsigned (&b_off)[kDwNPF], float (&b_keep)[kDwNPF], unsigned& b_own) __attribute__((always_inline)) {
            b_own = 0;
#pragma unroll
            for (int u = 0; u < kDwNPF; ++u) {
                const int rr = item_rr(u), hy = SH ? rr : 2 * pw + rr, hx = item_col(u);
                const int gy = tt.y0 + hy - 1, gx = tt.x0 + hx - 1;
                const bool in = item_id(u) < kItems && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
                b_off[u] = 4u * (in ? (unsigned)(((size_t)gy * a.W + gx) * a.Cin + w4 * 4) : (unsigned)(w4 * 4));
                b_keep[u] = in ? 1.f : 0.f;
                // the materialised input is written once per pixel: SH: every halo pixel has ONE loader, it owns the tile's own 8 x 16;
                // else the wave whose strip rows 1, 2 are the pixel's tile row
                const bool mine = SH ? (hy >= 1 && hy <= kPcTH) : (rr == 1 || rr == 2);
                if (in && mine && hx >= 1 && hx <= kPcTW) b_own |= 1u << u;
            }
        };
        struct Regs { f32x4 pre[kDwNPF]; f32x4 prer[RES ? kDwNPF : 1]; f32x4 ss[2]; f32x4 rs[2]; };
        DwTile tl = dw_decode(first, a), tn = tl;
        // raw words of one unit = (slice zrel of stage s, channel block of stage s) -> registers
        auto issue = [&](bool nx, int t, int cb, bool unitB, Regs& r) __attribute__((always_inline)) {
            r.rs[0] = r.rs[1] = f32x4{1.f, 1.f, 0.f, 0.f};
            if constexpr (!IDENT) {
                r.ss[0] = *reinterpret_cast<const f32x4*>(ssl + 2 * (cb * kCB + w4 * 4));
                r.ss[1] = *reinterpret_cast<const f32x4*>(ssl + 2 * (cb * kCB + w4 * 4) + 4);
            }
            if constexpr (RES && !RSID) {
                r.rs[0] = *reinterpret_cast<const f32x4*>(ssl + 2 * a.Cin + 2 * (cb * kCB + w4 * 4));
                r.rs[1] = *reinterpret_cast<const f32x4*>(ssl + 2 * a.Cin + 2 * (cb * kCB + w4 * 4) + 4);
            }
            const int tz = (nx ? tn.z0 : tl.z0) + (unitB ? dw_zB(t) : dw_zA(t));
            // (z, cb) are wave-uniform; saying so keeps the base in SGPRs and the loads in their saddr form (uniform 64-bit base +
            // 32-bit lane offset) — otherwise every load of the stage loop pays a v_lshl_add_u64
            const int z = __builtin_amdgcn_readfirstlane(min(max(tz, 0), a.N - 1));    // clamped: an outside slice is not used when published
            const size_t base = ((size_t)z * plane + (size_t)(__builtin_amdgcn_readfirstlane(cb) * kCB)) * sizeof(float);
            const __amdgpu_buffer_rsrc_t xb = pc_rsrc(reinterpret_cast<const char*>(a.x) + base);
            const __amdgpu_buffer_rsrc_t rb = pc_rsrc(reinterpret_cast<const char*>(RES ? a.res : a.x) + base);
#pragma unroll
            for (int u = 0; u < kDwNPF; ++u) {
                const unsigned o = nx ? nxt_off[u] : cur_off[u];
                r.pre[u] = pc_bload(xb, o);
                if constexpr (RES) r.prer[u] = pc_bload(rb, o);
            }
        };
        setup(tl, cur_off, cur_keep, cur_own);
        Regs setA, setB;
        issue(false, 0, 0, false, setA);
        issue(false, 0, 0, true, setB);
        int qbuf = 0;
        bool has_next = false;

        // normalise / activate one unit and publish it: COMBINE = false: strip = v;  true: strip = strip + sgn * v
        auto publish = [&](auto comb_tag, auto interior_tag, Regs& r, int z, int cb, bool wmat, float sgn) __attribute__((always_inline)) {
            constexpr bool COMBINE = decltype(comb_tag)::value;
            constexpr bool INTERIOR = decltype(interior_tag)::value;   // every item of every lane inside the image: no padding mask
            // the (scale, shift) words are re-paired for the packed FMAs HERE and not where they were loaded: without this the
            // compiler hoists the eight moves to right behind the loads, i.e. waits for the prefetch the moment it is issued
            if constexpr (!IDENT) asm volatile("" : "+v"(r.ss[0]), "+v"(r.ss[1]));
            if constexpr (RES && !RSID) asm volatile("" : "+v"(r.rs[0]), "+v"(r.rs[1]));
            const f32x2 sc01 = r.ss[0].lo, sh01 = r.ss[0].hi, sc23 = r.ss[1].lo, sh23 = r.ss[1].hi;   // the LDS table is stored pre-paired (pc_ss_slot)
            const f32x2 rc01 = r.rs[0].lo, rh01 = r.rs[0].hi, rc23 = r.rs[1].lo, rh23 = r.rs[1].hi;
            const f32x2 sg2 = {sgn, sgn};
            auto group = [&](auto u0_tag, auto u1_tag) __attribute__((always_inline)) {
                constexpr int U0 = decltype(u0_tag)::value, U1 = decltype(u1_tag)::value, NU = U1 - U0;
                f32x2 lo[NU], hi[NU];
                f32x4 old[COMBINE ? NU : 1];
                if constexpr (COMBINE) {
                    // the strip words unit A left: this lane wrote them itself (in-order LDS, no barrier); the memory clobber
                    // keeps the compiler from forwarding the stored registers across the refill instead (20 live VGPRs)
                    asm volatile("" ::: "memory");
#pragma unroll
                    for (int i = 0; i < NU; ++i) old[i] = *reinterpret_cast<const f32x4*>(raw + wr_off[U0 + i]);
                }
                if constexpr (IDENT) {
#pragma unroll
                    for (int i = 0; i < NU; ++i) { lo[i] = r.pre[U0 + i].lo; hi[i] = r.pre[U0 + i].hi; }
                } else if constexpr (CLAMP) {
#pragma unroll
                    for (int i = 0; i < NU; ++i) {
                        lo[i] = pk_fma_clamp01(r.pre[U0 + i].lo, sc01, sh01);
                        hi[i] = pk_fma_clamp01(r.pre[U0 + i].hi, sc23, sh23);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                } else {
#pragma unroll
                    for (int i = 0; i < NU; ++i) {
                        lo[i] = __builtin_elementwise_fma(r.pre[U0 + i].lo, sc01, sh01);
                        hi[i] = __builtin_elementwise_fma(r.pre[U0 + i].hi, sc23, sh23);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    if (a.x_relu) {
#pragma unroll
                        for (int i = 0; i < NU; ++i) { lo[i].x = relu1(lo[i].x); lo[i].y = relu1(lo[i].y); hi[i].x = relu1(hi[i].x); hi[i].y = relu1(hi[i].y); }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                if constexpr (RES && RSID) {
#pragma unroll
                    for (int i = 0; i < NU; ++i) { lo[i] = lo[i] + r.prer[U0 + i].lo; hi[i] = hi[i] + r.prer[U0 + i].hi; }
                    __builtin_amdgcn_sched_barrier(0);
                } else if constexpr (RES) {
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
                    f32x4 v = __builtin_shufflevector(lo[i], hi[i], 0, 1, 2, 3);
                    if (MAT && wmat) {   // the activated input is written once per slice: by the wave that owns the pixel, in phase 1
                        if ((cur_own >> u) & 1u)
                            *reinterpret_cast<f32x4*>(reinterpret_cast<char*>(a.mat) + ((size_t)z * plane + (size_t)(cb * kCB)) * sizeof(float) + cur_off[u]) = v;
                    }
                    if constexpr (COMBINE) {
                        const f32x2 cl = __builtin_elementwise_fma(lo[i], sg2, old[i].lo), ch = __builtin_elementwise_fma(hi[i], sg2, old[i].hi);
                        v = __builtin_shufflevector(cl, ch, 0, 1, 2, 3);
                    }
                    *reinterpret_cast<f32x4*>(raw + wr_off[u]) = v;
                }
            };
            if constexpr (RES && kDwNPF > 3) {
                group(std::integral_constant<int, 0>{}, std::integral_constant<int, 3>{});
                group(std::integral_constant<int, 3>{}, std::integral_constant<int, kDwNPF>{});
            } else {
                group(std::integral_constant<int, 0>{}, std::integral_constant<int, kDwNPF>{});
            }
        };

        // plane transform B^T d B of this lane's (tile, word): rows (2 of the 4 xi_y), then columns; strip rawT -> V[qbuf]
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
        int gi = 0;                            // iterations so far (SH: strip parity; the transform lags one iteration)

        for (int it = 0; it < count; ++it) {
            has_next = it + 1 < count;
            // a tile whose 10x18 halo lies inside the image needs no zero-padding mask (80 % of the tiles at config B)
            const bool interior = tl.y0 >= 1 && tl.y0 + kPcTH + 1 <= a.H && tl.x0 >= 1 && tl.x0 + kPcTW + 1 <= a.W;
            int cb = 0, t = 0;
            for (int s = 0; s < NS; ++s) {
                // the book of the next tile is needed by the refills of the tile's last stage
                if (s == NS - 1 && has_next) { tn = dw_decode(first + (it + 1) * step, a); setup(tn, nxt_off, nxt_keep, nxt_own); }
                if constexpr (SH) { raw = rawb + (gi & 1) * kDwShStrip; rawT = rawb + ((gi & 1) ^ 1) * kDwShStrip; }
                const int zA = tl.z0 + dw_zA(t), zB = tl.z0 + dw_zB(t);
                const bool zinA = zA >= 0, zinB = zB < a.N;       // zA <= z0 + 1 < N and zB >= z0 >= 0 always hold
                const bool wmat = MAT && t == 1 && tl.cg == 0;    // phase 1 publishes slices z0 (unit A) and z0 + 1 (unit B)
                const bool nx = s + 1 >= NS;
                const int cbn = cb + 1 == ncb ? 0 : cb + 1, tnx = nx ? 0 : (cb + 1 == ncb ? t + 1 : t);   // (t, position) of stage s + 1
                const int cbe = dw_cb(t, cb, ncb), cbne = dw_cb(tnx, cbn, ncb);                            // their channel blocks
                if constexpr (MAT) {
                    // With materialise stores in the queue the compiler cannot count it (loads and stores of one wave pending =
                    // "may complete out of order" = every wait becomes vmcnt(0)), and the wait for set B would also wait for the
                    // refill of set A issued just before it.  So BOTH sets are waited for here, at the one point of the stage where
                    // nothing else is in flight, and passed through an opaque asm: later uses no longer depend on the loads.
#pragma unroll
                    for (int u = 0; u < kDwNPF; ++u) {
                        asm volatile("" : "+v"(setA.pre[u]), "+v"(setB.pre[u]));
                        if constexpr (RES) asm volatile("" : "+v"(setA.prer[u]), "+v"(setB.prer[u]));
                    }
                    asm volatile("" : "+v"(setA.ss[0]), "+v"(setA.ss[1]), "+v"(setB.ss[0]), "+v"(setB.ss[1]));
                    if constexpr (RES && !RSID) asm volatile("" : "+v"(setA.rs[0]), "+v"(setA.rs[1]), "+v"(setB.rs[0]), "+v"(setB.rs[1]));
                }
                // (1) unit A -> strip
                if (abl & (2 | 8)) {
                } else if (!zinA) {
#pragma unroll
                    for (int u = 0; u < kDwNPF; ++u) *reinterpret_cast<f32x4*>(raw + wr_off[u]) = f32x4{0.f, 0.f, 0.f, 0.f};
                } else {
                    if (interior) publish(std::false_type{}, std::true_type{}, setA, zA, cbe, wmat, 1.f);
                    else publish(std::false_type{}, std::false_type{}, setA, zA, cbe, wmat, 1.f);
                }
                // refills are UNCONDITIONAL (the last stage of the last tile re-reads this tile's stage 0: harmless, never used);
                // a set is refilled right after it was published (longest time to land)
                if (!(abl & (2 | 32))) issue(nx && has_next, tnx, cbne, false, setA);
                // (2) unit B combined into the strip: V_t = d[zA] + sign d[zB], sign = +1 in phase 1 only
                if (zinB && !(abl & (2 | 16))) {
                    if (interior) publish(std::true_type{}, std::true_type{}, setB, zB, cbe, wmat, t == 1 ? 1.f : -1.f);
                    else publish(std::true_type{}, std::false_type{}, setB, zB, cbe, wmat, t == 1 ? 1.f : -1.f);
                }
                if (!(abl & (2 | 32))) issue(nx && has_next, tnx, cbne, true, setB);
                if constexpr (!SH) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // private strip: the transform reads what this wave just wrote
                // (3) plane transform B^T d B of this lane's (tile, word) — SH: of the stage published one iteration ago
                if (!(abl & (2 | 4)) && (!SH || gi > 0)) transform();
                __syncthreads();
                if (!SH || gi > 0) qbuf ^= 1;
                ++gi;
                if (++cb == ncb) { cb = 0; ++t; }
            }
            tl = tn;
#pragma unroll
            for (int u = 0; u < kDwNPF; ++u) { cur_off[u] = nxt_off[u]; cur_keep[u] = nxt_keep[u]; }
            cur_own = nxt_own;
        }
        if constexpr (SH) {                    // the last published stage
            rawT = rawb + ((gi & 1) ^ 1) * kDwShStrip;
            transform();
            __syncthreads();
        }
        __syncthreads();                       // the consumers' last stage
    }
}

// w [Cout][Cin][3][3][3] -> U_t = sum_kd Gd[t][kd] (G g_kd G^T) (float64, rounded once) in the kernel's B-operand order
// [cg][stage = t*ncb + cb][xi][wave][lane = kq*16 + j][e], co = cg*64 + 16*wave + j, ci = cb*16 + 4*kq + e
__global__ __launch_bounds__(256) void conv_wino_dw_pack_kernel(const float* __restrict__ w, float* __restrict__ wp, int Cin, int Cout,
                                                                int transposed) {
    const long total = (long)Cout * Cin * 4 * 16;
    if (transposed == 2) {     // both streams in one launch (grid.y = 2): forward at wp, data gradient behind it (see conv_wino_pack_kernel)
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
    const int stage = (int)(t % (4 * ncb));
    const int cg = (int)(t / (4 * ncb));
    const int td = stage / ncb, cb = stage - td * ncb;
    const int co = cg * 64 + 16 * wave + j, ci = cb * kCB + 4 * kq + e;
    const double G[4][3] = {{1.0, 0.0, 0.0}, {0.5, 0.5, 0.5}, {0.5, -0.5, 0.5}, {0.0, 0.0, 1.0}};
    const int aa = xi >> 2, bb = xi & 3;
    double u = 0.0;
#pragma unroll
    for (int kd = 0; kd < 3; ++kd) {
        // transposed: the stored tensor is [Cin][Cout][3][3][3] (this kernel's ci is ITS output channel), taps flipped in every dimension
        const float* g = transposed ? w + (((size_t)ci * Cout + co) * 3 + (2 - kd)) * 9 : w + (((size_t)co * Cin + ci) * 3 + kd) * 9;
        double u2 = 0.0;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) u2 += G[aa][ky] * (double)g[transposed ? (2 - ky) * 3 + (2 - kx) : ky * 3 + kx] * G[bb][kx];
        u += G[td][kd] * u2;
    }
    wp[idx] = (float)u;
}

}  // namespace nrgbd

extern "C" int nrgbd_conv_wino_dw_pack(const float* w, float* w_wino, int Cin, int Cout, int transposed, void* stream) {
    using namespace nrgbd;
    if (!w || !w_wino) return NRGBD_E_NULL;
    if (Cin <= 0 || Cin % kCB || Cout <= 0 || Cout % 64) return NRGBD_E_SHAPE;
    if (transposed < 0 || transposed > 2) return NRGBD_E_ARG;
    if (transposed == 2 && Cin % 64) return NRGBD_E_SHAPE;
    const long total = (long)Cout * Cin * 4 * 16;
    hipLaunchKernelGGL(conv_wino_dw_pack_kernel, dim3((unsigned)((total + 255) / 256), transposed == 2 ? 2 : 1), dim3(256), 0, (hipStream_t)stream,
                       w, w_wino, Cin, Cout, transposed);
    NRGBD_CHECK_LAUNCH();
    return NRGBD_OK;
}

static int dw_workgroups(int N, int H, int W, int Cout, int* out) {
    const long nt = (long)(nrgbd_conv_wino_tiles(N, H, W, 1) / 2) * (Cout / 64);
    int dev = 0, ncu = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e == hipSuccess) e = hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev);
    if (e != hipSuccess) return (int)e;
    if (ncu <= 0) return NRGBD_E_ARG;
    *out = nt < ncu ? (int)nt : ncu;
    return NRGBD_OK;
}

extern "C" int nrgbd_conv_wino_dw_workgroups(int N, int H, int W, int Cout) {
    if (N <= 0 || (N & 1) || H <= 0 || W <= 0 || Cout <= 0 || Cout % 64) return NRGBD_E_SHAPE;
    int n = 0;
    const int rc = dw_workgroups(N, H, W, Cout, &n);
    return rc == NRGBD_OK ? n : (rc < 0 ? rc : NRGBD_E_ARG);
}

static int conv_wino_dw_launch(const float* x, const float* x_ss, int x_relu, const float* res, const float* res_ss,
                               int res_relu, float* materialized, const float* w_wino, float* y, float* stats, int N,
                               int H, int W, int Cin, int Cout, void* stream, float x_unit = 0.f) {
    using namespace nrgbd;
    if (!x || !w_wino || !y) return NRGBD_E_NULL;
    if (N <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cin % kCB || Cin > kDwMaxCin || Cout <= 0 || Cout % 64) return NRGBD_E_SHAPE;
    if (N & 1) return NRGBD_E_SHAPE;                                  // pairs of output slices
    if (H % kPcTH || W % kPcTW) return NRGBD_E_SHAPE;                 // whole 8x16 tiles only (every grid of the path; others: nrgbd_conv_wino_f32)
    if ((long)H * W * Cin >= (1L << 30)) return NRGBD_E_SHAPE;       // 32-bit BYTE offsets inside a slice (the slice is a 64-bit base)
    const int rows = nrgbd_conv_wino_tiles(N, H, W, 1);              // statistics rows: one per (8x16 tile, slice) as wino_pc
    const long nt = (long)(rows / 2) * (Cout / 64);
    if (nt >= (1L << 31)) return NRGBD_E_SHAPE;
    WinoPcArgs a{x, x_ss, res, res_ss, materialized, w_wino, y, stats, x_relu, res_relu, N, H, W, Cin, Cout, (int)nt, rows,
                 nullptr, 0, 0, 0, 0, dev_env_int("NRGBD_WINO_ABL"), x_unit};
    int dev = 0, ncu = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e == hipSuccess) e = hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev);
    if (e != hipSuccess) return (int)e;
    if (ncu <= 0) return NRGBD_E_ARG;
    const int nwg = nt < ncu ? (int)nt : ncu;   // persistent: one workgroup per CU
    const size_t lds = (size_t)(kDwNBuf * kPcV + kDwStrips + 4 * kDwStashWave + 4 * Cin) * sizeof(float);   // 64 + 25.6 (20) + 64 KB + tables
    // the function's opt-in is set to the form's maximum, not to this call's size (see nrgbd_conv_wino_f32: hipGraph replays read it)
    const int lds_attr = 160 * 1024;
    hipStream_t st = (hipStream_t)stream;
#define NRGBD_WINO_DW_LAUNCH(RES_, MAT_, RSID_)                                                                               \
    do {                                                                                                                      \
        e = set_max_dynamic_lds(reinterpret_cast<const void*>(&conv_wino_dw_kernel<RES_, MAT_, RSID_>),                       \
                                lds_attr);                                        \
        if (e != hipSuccess) return (int)e;                                                                                   \
        hipLaunchKernelGGL((conv_wino_dw_kernel<RES_, MAT_, RSID_>), dim3(nwg), dim3(512), lds, st, a);                        \
    } while (0)
    const bool rsid = res && !res_ss && !res_relu;
    if (x_unit != 0.f) {            // the CLAMP instantiation (nrgbd_conv_wino_dw_unit_f32 checked its preconditions)
        e = set_max_dynamic_lds(reinterpret_cast<const void*>(&conv_wino_dw_kernel<false, false, false, false, true>), lds_attr);
        if (e != hipSuccess) return (int)e;
        hipLaunchKernelGGL((conv_wino_dw_kernel<false, false, false, false, true>), dim3(nwg), dim3(512), lds, st, a);
    } else if (res && rsid) { if (materialized) NRGBD_WINO_DW_LAUNCH(true, true, true); else NRGBD_WINO_DW_LAUNCH(true, false, true); }
    else if (res) { if (materialized) NRGBD_WINO_DW_LAUNCH(true, true, false); else NRGBD_WINO_DW_LAUNCH(true, false, false); }
    else if (materialized) NRGBD_WINO_DW_LAUNCH(false, true, false);
    else if (NRGBD_DW_IDENT && !x_ss && !x_relu) {   // the IDENT instantiation: nothing to apply to x
        e = set_max_dynamic_lds(reinterpret_cast<const void*>(&conv_wino_dw_kernel<false, false, false, true>), lds_attr);
        if (e != hipSuccess) return (int)e;
        hipLaunchKernelGGL((conv_wino_dw_kernel<false, false, false, true>), dim3(nwg), dim3(512), lds, st, a);
    } else NRGBD_WINO_DW_LAUNCH(false, false, false);
#undef NRGBD_WINO_DW_LAUNCH
    NRGBD_CHECK_LAUNCH();
    return NRGBD_OK;
}

extern "C" int nrgbd_conv_wino_dw_f32(const float* x, const float* x_ss, int x_relu, const float* res, const float* res_ss,
                                      int res_relu, float* materialized, const float* w_wino, float* y, float* stats, int N,
                                      int H, int W, int Cin, int Cout, void* stream) {
    return conv_wino_dw_launch(x, x_ss, x_relu, res, res_ss, res_relu, materialized, w_wino, y, stats, N, H, W, Cin, Cout, stream);
}

// The plain form with relu(x * s + t) as a clamped FMA (the CLAMP instantiation): x_unit = 2^-k, the weight stream packed from
// 2^k * w; see include/nrgbd.h.
extern "C" int nrgbd_conv_wino_dw_unit_f32(const float* x, const float* x_ss, float x_unit, const float* w_wino, float* y, float* stats,
                                           int N, int H, int W, int Cin, int Cout, void* stream) {
    if (!x_ss) return NRGBD_E_NULL;
    int ex = 0;
    if (!(x_unit > 0.f) || x_unit > 1.f || frexpf(x_unit, &ex) != 0.5f) return NRGBD_E_ARG;   // a power of two in (0, 1]
    return conv_wino_dw_launch(x, x_ss, 1, nullptr, nullptr, 0, nullptr, w_wino, y, stats, N, H, W, Cin, Cout, stream, x_unit);
}
