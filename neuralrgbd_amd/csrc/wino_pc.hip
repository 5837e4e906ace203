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
//   consumer wave c (0..3) = output channels 16c .. 16c+15 of ALL 16 transform points and all 32 tiles of the workgroup's
//       8x16-pixel tile (128 accumulator VGPRs): per stage 16 x (2 LDS reads + 1 weight line + 8 v_mfma_f32_16x16x4_f32);
//       weight lines (1 KB, packed per wave) run 7 steps ahead in an 8-deep register ring that continues across stages and
//       tiles; the first A operand of the next stage is read before the stage barrier (three V buffers make that legal).
//   producer wave p (0..3) = tile row p (8 Winograd tiles): raw rows 2p..2p+3 of the 10x18 halo of one 16-channel block
//       global -> registers (issued one stage earlier) -> BatchNorm / ReLU / residual -> its PRIVATE 5 KB strip of LDS (no
//       workgroup barrier: a wave's LDS operations execute in order) -> B^T d B per (tile, 16-byte word, half) -> V[q % 3].
//   One s_barrier per stage; the consumers never wait for data (producers are two stages ahead), the producers wait for
//   the consumers — which is the point: the matrix pipe is the resource to keep busy.
//   Persistent: one workgroup per CU walks its share of the tile list (XCD-aware: an XCD's workgroups sweep neighbouring
//   tiles, depth fastest, so the three slices a 3-D tile needs are shared in that XCD's L2), so a tile's epilogue and the
//   next tile's first loads overlap with the producers' run-ahead instead of being exposed at every workgroup boundary.
// LDS: 3 x 32 KB V + 4 x 5 KB raw strips = 116 KB (one workgroup per CU; 2 waves per SIMD, up to 256 VGPRs each).
#include "conv_tile.hpp"

namespace nrgbd {

constexpr int kPcTH = 8, kPcTW = 16;            // output pixels of a tile, in units of the dilation lattice
constexpr int kPcTiles = 32;                    // Winograd tiles per workgroup tile: ty = tile >> 3, tx = tile & 7
constexpr int kPcRawW = 20;                     // raw strip row pitch in pixels (18 used; a multiple of 4 keeps a row's bank map)
constexpr int kPcRawWave = 4 * kPcRawW * kCB;   // floats of one producer wave's 4-row strip (5 KB)
constexpr int kPcV = 16 * kPcTiles * kCB;       // floats of one V buffer [16 xi][32 tiles][16] (32 KB)
constexpr int kPcNBuf = 3;
constexpr int kPcItems = 4 * 18 * 4;            // (row, column, 16-byte word) items of a strip
constexpr int kPcNPF = (kPcItems + 63) / 64;    // per producer lane and stage (5)
constexpr int kPcBD = 7, kPcNB = 8;             // weight ring: distance / slots

struct WinoPcArgs {
    const float* x;       // [N][H][W][Cin] raw input (pre-activation); N = depth slices when KD = 3
    const float* x_ss;    // [Cin][2] (scale, shift) applied to x, or null
    const float* res;     // second operand added after activation, or null
    const float* res_ss;  // [Cin][2] for res, or null
    float* mat;           // materialised input act(x) + act(res), or null
    const float* wp;      // Winograd-domain weights [Cout/64][stage = cb*KD + kd][16 xi][4 waves][64 lanes][4]
    float* y;             // [N][H][W][Cout] raw convolution output
    float* stats;         // [spatial tiles][2*Cout]: per-channel sum and sum of squares of y, or null
    int x_relu, res_relu;
    int N, H, W, Cin, Cout;
    int ntiles;           // spatial tiles x Cout/64
};

struct PcTile { int n, y0, x0, py, px, cg, row; };

template <int KD, int DIL>
__device__ __forceinline__ PcTile pc_decode(int t, const WinoPcArgs& a) {
    PcTile r;
    const int ncg = a.Cout >> 6;
    const int tiles_x = (a.W + kPcTW * DIL - 1) / (kPcTW * DIL), tiles_y = (a.H + kPcTH * DIL - 1) / (kPcTH * DIL);
    r.row = t / ncg;
    r.cg = t - r.row * ncg;
    t = r.row;
    r.n = 0;
    if (KD == 3) { r.n = t % a.N; t /= a.N; }   // depth fastest: the three workgroups that read one slice are list neighbours
    int par = 0;
    if (DIL > 1) { par = t % (DIL * DIL); t /= DIL * DIL; }
    const int tx = t % tiles_x; t /= tiles_x;
    int ty = t;
    if (KD != 3) { ty = t % tiles_y; r.n = t / tiles_y; }
    r.py = par / DIL; r.px = par - r.py * DIL;
    r.y0 = ty * kPcTH * DIL; r.x0 = tx * kPcTW * DIL;
    return r;
}

// LDS image of V: [xi*32 + tile][16 floats]; the 16-byte slot s of a tile is stored at slot (s + 2*((tile >> 3) & 1)) & 3, so
// that every ds_read_b128 service group of the 16x16x4 A-operand pattern hits 16 different bank quads (as in generation 1)
__device__ __forceinline__ int pc_slot(int xi, int tile, int slot) {
    return ((xi * kPcTiles + tile) << 4) + (((slot + 2 * ((tile >> 3) & 1)) & 3) << 2);
}

template <int KD, int DIL, bool RES>
__global__ __launch_bounds__(512) void conv_wino_pc_kernel(const WinoPcArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* Vb = lds;                           // [3][16 xi][32 tiles][16]
    float* rawb = lds + kPcNBuf * kPcV;        // [4 producer waves][4 rows][20 pixels][16]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int NS = (a.Cin / kCB) * KD;         // stages per tile

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

    if (wv < 4) {
        // =========================================== consumer: 16 output channels x 16 xi x 32 tiles ====================
        const int kq = lane >> 4, jj = lane & 15;
        f32x4 acc[16][2];
#pragma unroll
        for (int xi = 0; xi < 16; ++xi) { acc[xi][0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[xi][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }
        const int a0 = pc_slot(0, jj, kq), a1 = pc_slot(0, 16 + jj, kq);   // + xi * 512 floats + buffer
        const f32x4* wbase = reinterpret_cast<const f32x4*>(a.wp) + wv * 64 + lane;
        const size_t wgroup = (size_t)NS * 16 * 256;                        // f32x4 per 64-column output group

        PcTile tl = pc_decode<KD, DIL>(first, a);
        const f32x4* wt = wbase + (size_t)tl.cg * wgroup;
        f32x4 Bn[kPcNB], An[2][2];
#pragma unroll
        for (int b = 0; b < kPcBD; ++b) Bn[b] = wt[b * 256];
        __syncthreads();                       // producers finish stage 0
        __syncthreads();                       // ... and stage 1
        An[0][0] = *reinterpret_cast<const f32x4*>(Vb + a0);
        An[0][1] = *reinterpret_cast<const f32x4*>(Vb + a1);
        int buf = 0;
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
#pragma unroll
                for (int xi = 0; xi < 16; ++xi) {
                    const int cur = xi & 1, nxt = cur ^ 1;
                    if (xi + 1 < 16) {
                        An[nxt][0] = *reinterpret_cast<const f32x4*>(Vc + a0 + (xi + 1) * (kPcTiles * kCB));
                        An[nxt][1] = *reinterpret_cast<const f32x4*>(Vc + a1 + (xi + 1) * (kPcTiles * kCB));
                    } else {   // first operand of the next stage: its buffer was completed two barriers ago
                        An[nxt][0] = *reinterpret_cast<const f32x4*>(Vn + a0);
                        An[nxt][1] = *reinterpret_cast<const f32x4*>(Vn + a1);
                    }
                    Bn[(xi + kPcBD) % kPcNB] = xi + kPcBD < 16 ? wcur[(xi + kPcBD) * 256] : wnx[(xi + kPcBD - 16) * 256];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        acc[xi][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(An[cur][0][e], Bn[xi % kPcNB][e], acc[xi][0], 0, 0, 0);
                        acc[xi][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(An[cur][1][e], Bn[xi % kPcNB][e], acc[xi][1], 0, 0, 0);
                        // pin the order: the two row blocks alternate (no back-to-back dependent MFMAs) and the operand
                        // streams keep their distances
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
                __syncthreads();
                buf = nbuf;
            }
            // ---- inverse transform Y = A^T M A in registers + output + per-channel partial statistics ----
            // lane (kq, jj): output channel co = 16 wv + jj; register r of row block m = tile 16 m + 4 kq + r
            const int co = tl.cg * 64 + wv * 16 + jj;
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int m = 0; m < 2; ++m) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int tile = 16 * m + 4 * kq + r;
                    const int ty = tile >> 3, tx = tile & 7;
                    float tr[2][4];   // t[a][xi_x] = sum_xi_y A^T[a][xi_y] M[xi_y][xi_x]
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
                        const int gy = tl.y0 + tl.py + DIL * (2 * ty + aa), gx = tl.x0 + tl.px + DIL * (2 * tx);
                        if (gy < a.H) {
                            float* o = a.y + (((size_t)tl.n * a.H + gy) * a.W + gx) * a.Cout + co;
                            if (gx < a.W) { o[0] = o0; s1 += o0; s2 = __builtin_fmaf(o0, o0, s2); }
                            if (gx + DIL < a.W) { o[(size_t)DIL * a.Cout] = o1; s1 += o1; s2 = __builtin_fmaf(o1, o1, s2); }
                        }
                    }
                }
            }
            if (a.stats) {   // the wave owns its 16 channels: reduce over the 4 lanes (kq) that share a channel
                s1 += __shfl_xor(s1, 16, 64); s2 += __shfl_xor(s2, 16, 64);
                s1 += __shfl_xor(s1, 32, 64); s2 += __shfl_xor(s2, 32, 64);
                if (kq == 0) {
                    a.stats[(size_t)tl.row * (2 * a.Cout) + co] = s1;
                    a.stats[(size_t)tl.row * (2 * a.Cout) + a.Cout + co] = s2;
                }
            }
#pragma unroll
            for (int xi = 0; xi < 16; ++xi) { acc[xi][0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[xi][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }
            tl = tn;
            wt = wt_next;
        }
    } else {
        // =========================================== producer: tile row pw (8 Winograd tiles) ===========================
        const int pw = wv - 4;
        float* raw = rawb + pw * kPcRawWave;
        const int w4 = lane & 3;
        // load / publish items: item = lane + 64u -> strip pixel pi = item >> 2 in (row, de-interleaved column) order
        int it_rr[kPcNPF], it_col[kPcNPF], wr_off[kPcNPF];
#pragma unroll
        for (int u = 0; u < kPcNPF; ++u) {
            const int pi = (lane + 64 * u) >> 2;
            const int rr = pi / 18, cp = pi - rr * 18;
            it_rr[u] = rr;
            it_col[u] = cp < 9 ? 2 * cp : 2 * cp - 17;      // strip columns stored even ones first, then odd ones
            wr_off[u] = (rr * kPcRawW + cp) * kCB + w4 * 4;
        }
        // transform item of this lane: (tile of the row, 16-byte word, half of the xi rows); 16 consecutive lanes = 4 tiles x 4
        // words: conflict-free strip reads (the four tiles' columns are consecutive strip pixels) and V writes
        const int tword = lane & 3, txl = ((lane >> 5) << 2) | ((lane >> 2) & 3), thalf = (lane >> 4) & 1;
        const int ttile = pw * 8 + txl;
        const int rd0 = (thalf * kPcRawW + txl) * kCB + tword * 4;   // strip (row thalf, column 2 txl): cc = 0; cc=1: +9 px; 2: +1; 3: +10

        unsigned off[kPcNPF], ok = 0, own = 0;
        PcTile tl = pc_decode<KD, DIL>(first, a);
        auto setup = [&](const PcTile& t) {
            ok = 0; own = 0;
#pragma unroll
            for (int u = 0; u < kPcNPF; ++u) {
                const int hy = 2 * pw + it_rr[u], hx = it_col[u];
                const int gy = t.y0 + t.py + DIL * (hy - 1), gx = t.x0 + t.px + DIL * (hx - 1);
                const bool in = (lane + 64 * u) < kPcItems && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
                const unsigned n2 = KD == 3 ? 0u : (unsigned)t.n;
                off[u] = in ? (unsigned)((((size_t)n2 * a.H + gy) * a.W + gx) * a.Cin + w4 * 4) : (unsigned)(w4 * 4);
                if (in) ok |= 1u << u;
                if (in && (it_rr[u] == 1 || it_rr[u] == 2) && hx >= 1 && hx <= kPcTW) own |= 1u << u;
            }
        };
        f32x4 pre[kPcNPF], prer[RES ? kPcNPF : 1];
        auto issue = [&](const PcTile& t, int s) {   // raw words of stage s of tile t -> registers
            const int cb = s / KD, kd = s - cb * KD;
            const int z = KD == 3 ? min(max(t.n + kd - 1, 0), a.N - 1) : 0;   // clamped: an outside slice is zeroed when published
            const unsigned base = (unsigned)z * plane + (unsigned)(cb * kCB);
#pragma unroll
            for (int u = 0; u < kPcNPF; ++u) {
                pre[u] = *reinterpret_cast<const f32x4*>(a.x + base + off[u]);
                if constexpr (RES) prer[u] = *reinterpret_cast<const f32x4*>(a.res + base + off[u]);
            }
        };
        setup(tl);
        issue(tl, 0);
        int qbuf = 0;
        for (int it = 0; it < count; ++it) {
            for (int s = 0; s < NS; ++s) {
                const int cb = s / KD, kd = s - cb * KD;
                const int z = KD == 3 ? tl.n + kd - 1 : tl.n;
                const bool zin = KD != 3 || (z >= 0 && z < a.N);
                {   // (1) normalise / activate the prefetched words and publish them to this wave's strip
                    const int c = cb * kCB + w4 * 4;
                    float ss[8] = {1.f, 0.f, 1.f, 0.f, 1.f, 0.f, 1.f, 0.f}, rs[8] = {1.f, 0.f, 1.f, 0.f, 1.f, 0.f, 1.f, 0.f};
                    if (a.x_ss) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) ss[e] = a.x_ss[2 * c + e];
                    }
                    if (RES && a.res_ss) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) rs[e] = a.res_ss[2 * c + e];
                    }
                    const bool wmat = a.mat && (KD != 3 || kd == 1) && tl.cg == 0;
#pragma unroll
                    for (int u = 0; u < kPcNPF; ++u) {
                        if (lane + 64 * u >= kPcItems) continue;
                        f32x4 v = {0.f, 0.f, 0.f, 0.f};
                        if (zin && ((ok >> u) & 1u)) {   // zero padding applies to the ACTIVATED tensor
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
                            // the activated input is written once: by the wave that owns the pixel, at the centre tap
                            if (wmat && ((own >> u) & 1u))
                                *reinterpret_cast<f32x4*>(a.mat + (unsigned)z * (KD == 3 ? plane : 0u) + (unsigned)(cb * kCB) + off[u]) = v;
                        }
                        *reinterpret_cast<f32x4*>(raw + wr_off[u]) = v;
                    }
                }
                // (2) the next stage's words fly while this one is transformed (and while the consumers work through two more)
                if (s + 1 < NS) issue(tl, s + 1);
                else if (it + 1 < count) { tl = pc_decode<KD, DIL>(first + (it + 1) * step, a); setup(tl); issue(tl, 0); }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the strip is wave-private: in-order LDS, no barrier
                {   // (3) input transform B^T d B of this lane's (tile, word): rows first (2 of the 4 xi_y), then columns
                    f32x4 r0[4], r1[4], r2[4];
#pragma unroll
                    for (int cc = 0; cc < 4; ++cc) {
                        const int co = ((cc & 1) * 9 + (cc >> 1)) * kCB;
                        r0[cc] = *reinterpret_cast<const f32x4*>(raw + rd0 + co);
                        r1[cc] = *reinterpret_cast<const f32x4*>(raw + rd0 + kPcRawW * kCB + co);
                        r2[cc] = *reinterpret_cast<const f32x4*>(raw + rd0 + 2 * kPcRawW * kCB + co);
                    }
                    // half 0: strip rows 0,1,2 = patch rows d0,d1,d2 -> xi_y 0 = d0 - d2, xi_y 1 = d1 + d2
                    // half 1: strip rows 1,2,3 = patch rows d1,d2,d3 -> xi_y 2 = d2 - d1, xi_y 3 = d1 - d3
                    f32x4 ya[4], yb[4];
#pragma unroll
                    for (int cc = 0; cc < 4; ++cc) {
                        ya[cc] = thalf ? (r1[cc] - r0[cc]) : (r0[cc] - r2[cc]);
                        yb[cc] = thalf ? (r0[cc] - r2[cc]) : (r1[cc] + r2[cc]);
                    }
                    float* Vq = Vb + qbuf * kPcV;
                    const int xa = (2 * thalf) * 4, xb = (2 * thalf + 1) * 4;
                    *reinterpret_cast<f32x4*>(Vq + pc_slot(xa + 0, ttile, tword)) = ya[0] - ya[2];
                    *reinterpret_cast<f32x4*>(Vq + pc_slot(xa + 1, ttile, tword)) = ya[1] + ya[2];
                    *reinterpret_cast<f32x4*>(Vq + pc_slot(xa + 2, ttile, tword)) = ya[2] - ya[1];
                    *reinterpret_cast<f32x4*>(Vq + pc_slot(xa + 3, ttile, tword)) = ya[1] - ya[3];
                    *reinterpret_cast<f32x4*>(Vq + pc_slot(xb + 0, ttile, tword)) = yb[0] - yb[2];
                    *reinterpret_cast<f32x4*>(Vq + pc_slot(xb + 1, ttile, tword)) = yb[1] + yb[2];
                    *reinterpret_cast<f32x4*>(Vq + pc_slot(xb + 2, ttile, tword)) = yb[2] - yb[1];
                    *reinterpret_cast<f32x4*>(Vq + pc_slot(xb + 3, ttile, tword)) = yb[1] - yb[3];
                }
                __syncthreads();
                qbuf = qbuf == kPcNBuf - 1 ? 0 : qbuf + 1;
            }
        }
        __syncthreads();                       // the consumers' last two stages
        __syncthreads();
    }
}

}  // namespace nrgbd

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
    if (N <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cin % kCB || Cout <= 0 || Cout % 64) return NRGBD_E_SHAPE;
    if ((kd != 1 && kd != 3) || (dilation != 1 && dilation != 2) || (kd == 3 && dilation != 1)) return NRGBD_E_ARG;
    if ((long)N * H * W * Cin >= (1L << 32)) return NRGBD_E_SHAPE;   // 32-bit element offsets in the loader
    const int rows = nrgbd_conv_wino_tiles(N, H, W, dilation);
    const long nt = (long)rows * (Cout / 64);
    if (nt >= (1L << 31)) return NRGBD_E_SHAPE;
    WinoPcArgs a{x, x_ss, res, res_ss, materialized, w_wino, y, stats, x_relu, res_relu, N, H, W, Cin, Cout, (int)nt};
    int dev = 0, ncu = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e == hipSuccess) e = hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev);
    if (e != hipSuccess) return (int)e;
    if (ncu <= 0) return NRGBD_E_ARG;
    const int nwg = nt < ncu ? (int)nt : ncu;   // persistent: one workgroup per CU
    const size_t lds = (size_t)(kPcNBuf * kPcV + 4 * kPcRawWave) * sizeof(float);   // 96 KB V + 20 KB strips
    hipStream_t st = (hipStream_t)stream;
#define NRGBD_WINO_PC_LAUNCH(KD_, DIL_, RES_)                                                                       \
    do {                                                                                                            \
        /* > 64 KB of dynamic LDS needs the opt-in; idempotent and ~1 us, so simply repeated per call (re-entrant) */ \
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wino_pc_kernel<KD_, DIL_, RES_>),               \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                              \
        if (e != hipSuccess) return (int)e;                                                                         \
        hipLaunchKernelGGL((conv_wino_pc_kernel<KD_, DIL_, RES_>), dim3(nwg), dim3(512), lds, st, a);               \
    } while (0)
    if (kd == 3) {
        if (res) NRGBD_WINO_PC_LAUNCH(3, 1, true); else NRGBD_WINO_PC_LAUNCH(3, 1, false);
    } else if (dilation == 1) {
        if (res) NRGBD_WINO_PC_LAUNCH(1, 1, true); else NRGBD_WINO_PC_LAUNCH(1, 1, false);
    } else {
        if (res) NRGBD_WINO_PC_LAUNCH(1, 2, true); else NRGBD_WINO_PC_LAUNCH(1, 2, false);
    }
#undef NRGBD_WINO_PC_LAUNCH
    NRGBD_CHECK_LAUNCH();
    return NRGBD_OK;
}
