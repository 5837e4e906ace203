// costvol_lds.hip — generation 2 of the fused plane-sweep cost volume: source texels staged in LDS.
//
// Why: the generation-1 kernel gathers 16-byte words straight from HBM/L2; 64 lanes hit 64 different
// cache lines per instruction and the texture-address path retires one lane per clock
// (1.39 ms at the 192x256x64 grid = exactly 64 clk per wave-load).  The LDS services a wave's
// ds_read_b128 in 4 clocks (256 B/clk/CU), 16x faster, provided the texel stride is an ODD number of
// 16-byte words (then the 16 lanes of a b128 lane group fall on 16 different bank quads).
//
// Work decomposition
//   workgroup (256 threads)  = one 16x16 tile of reference pixels x one group of 8 consecutive depth
//                              candidates (grid = tiles x ceil(D/8));
//   thread                   = one reference pixel, its 68-channel texel held in registers;
//   for each source view: the tile's footprint in the source image — the bounding box of the
//   bilinear taps of the 4 tile-corner pixels, which bounds every pixel of the tile because a
//   homography maps the (convex) tile onto a convex quadrilateral — is copied HBM -> LDS one channel
//   block at a time ([texel][9 x 16 B], stride 144 B), shared by as many of the 8 candidates as fit
//   the 63 KB patch (far planes move the footprint by < 1 texel per candidate, so usually all 8).
//   Candidates whose footprint does not fit even alone (zoom > ~1.1) fall back to a direct gather by all threads.
// Two workgroups are resident per CU so one stages while the other computes.
//
// Arithmetic is identical to generation 1 (same helpers), so both satisfy the same parity tests.
#include "costvol.hpp"

namespace nrgbd {

constexpr int kTile = 16;       // tile edge (pixels)
constexpr int kKG = 8;          // depth candidates per workgroup
constexpr int kSingles = 8;     // leading candidates that may get a workgroup each (small grids)
constexpr int kPatchF4 = 4032;  // float4 slots of the source patch (63 KB)

// Channel blocking: a texel of CP4 16-byte words is processed in NCB blocks of up to 9 words.  Every
// block STAGES exactly NS = min(CP4, 9) words starting at word W0(cb) = min(9 cb, CP4 - NS) (the last
// block overlaps its predecessor instead of being short), so the LDS image of a block is the plain
// linear array [texel][NS] whenever NS is odd — stride NS | 1 words = an odd number of 16-B words,
// which spreads the 16 lanes of a ds_read_b128 lane group over 16 different bank quads.
constexpr int kKB = 9;          // 16-byte words of a texel per channel block
template <int CP4>
struct LdsCfg {
    static constexpr int NCB = (CP4 + kKB - 1) / kKB;
    static constexpr int NS = CP4 < kKB ? CP4 : kKB;   // words staged per texel per block
    static constexpr int S4 = NS | 1;              // LDS texel stride in 16-B words (odd)
    static constexpr int PMAX = kPatchF4 / S4;     // texels that fit
    static constexpr int MAXIT = (kPatchF4 + 255) / 256;
    static constexpr int w0(int cb) { return (cb * kKB < CP4 - NS) ? cb * kKB : CP4 - NS; }
    static constexpr int first(int cb) { return cb * kKB; }                                      // first word computed
    static constexpr int count(int cb) { return (CP4 - cb * kKB < kKB) ? CP4 - cb * kKB : kKB; }  // words computed
};

template <int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (N > 0) {
        static_for<N - 1>(f);
        f(std::integral_constant<int, N - 1>{});
    }
}

// Direct-gather evaluation of one (pixel, candidate, view): sum_c dist(sample_c, ref_c) with 16-byte
// loads straight from HBM/L2: the path of candidates whose tile footprint overflows the patch even alone.
__device__ __noinline__ float gather_point(const float* __restrict__ sv, const float* __restrict__ refp,
                                           float ix, float iy, int w, int h, int Cp, int C, int dist) {
    const Bilinear b = bilinear_zeros(ix, iy, w, h);
    const float4* rp0 = reinterpret_cast<const float4*>(refp);
    if (b.nw == 0.f && b.ne == 0.f && b.sw == 0.f && b.se == 0.f) {   // all four taps out of view: distance to the zero vector
        float acc0 = 0.f;
        for (int i = 0; i < (Cp >> 2); ++i) {
            const float4 rr = rp0[i];
            const float s[4] = {0.f - rr.x, 0.f - rr.y, 0.f - rr.z, 0.f - rr.w};
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (4 * i + e < C) acc0 = (dist == NRGBD_DIST_L2) ? __builtin_fmaf(s[e], s[e], acc0) : acc0 + fabsf(s[e]);
        }
        return acc0;
    }
    const float4* pnw = reinterpret_cast<const float4*>(sv + ((size_t)b.y0 * w + b.x0) * Cp);
    const float4* pne = reinterpret_cast<const float4*>(sv + ((size_t)b.y0 * w + b.x1) * Cp);
    const float4* psw = reinterpret_cast<const float4*>(sv + ((size_t)b.y1 * w + b.x0) * Cp);
    const float4* pse = reinterpret_cast<const float4*>(sv + ((size_t)b.y1 * w + b.x1) * Cp);
    const float4* rp = reinterpret_cast<const float4*>(refp);
    float acc = 0.f;
    for (int i = 0; i < (Cp >> 2); ++i) {
        const float4 A = pnw[i], B = pne[i], Cc = psw[i], Dd = pse[i], rr = rp[i];
        const float s[4] = {lerp4(A.x, B.x, Cc.x, Dd.x, b) - rr.x, lerp4(A.y, B.y, Cc.y, Dd.y, b) - rr.y,
                            lerp4(A.z, B.z, Cc.z, Dd.z, b) - rr.z, lerp4(A.w, B.w, Cc.w, Dd.w, b) - rr.w};
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (4 * i + e < C) acc = (dist == NRGBD_DIST_L2) ? __builtin_fmaf(s[e], s[e], acc) : acc + fabsf(s[e]);
    }
    return acc;
}

// Footprint in the source image of the pixel rectangle [xa,xb] x [ya,yb] of the reference on the plane
// at depth dc: bounding box of the bilinear taps of the 4 corner pixels (+1 texel of slack), clipped to
// the image.  A homography maps the rectangle onto a convex quadrilateral, so the corners bound every
// pixel inside as long as the plane stays in front of the source camera (den > 0 at the 4 corners =>
// den > 0 inside, den being affine in the pixel position).  Returns 0 = nothing in view, 1 = box valid,
// 2 = unbounded (plane crosses the camera): caller must gather.
struct Box { int xlo, xhi, ylo, yhi; };
__device__ __forceinline__ int region_box(const CostvolArgs& a, const float* KRv, const float* Ktv, float dc,
                                          int xa, int xb, int ya, int yb, Box& o) {
    const size_t hw = (size_t)a.h * a.w;
    const float wf = (float)a.w, hf = (float)a.h;
    const int cxs[2] = {xa, xb}, cys[2] = {ya, yb};
    float mnx = INFINITY, mxx = -INFINITY, mny = INFINITY, mxy = -INFINITY;
    bool ok = true;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const size_t pc = (size_t)cys[c >> 1] * a.w + cxs[c & 1];
        const SweepTerm sc = make_sweep_term(KRv, Ktv, a.rays[pc], a.rays[hw + pc], a.rays[2 * hw + pc]);
        const float den = (sc.t1z + sc.t2z * dc) + 1e-10f;
        float ix, iy;
        sweep_sample_pos(sc, dc, a.cx, a.cy, wf, hf, a.align != 0, ix, iy);
        ok = ok && (den > 0.f) && (fabsf(ix) < 1e8f) && (fabsf(iy) < 1e8f);
        mnx = fminf(mnx, ix); mxx = fmaxf(mxx, ix);
        mny = fminf(mny, iy); mxy = fmaxf(mxy, iy);
    }
    if (!ok) { o = Box{0, 1 << 20, 0, 1 << 20}; return 2; }
    mnx = fmaxf(mnx, -4.f); mxx = fminf(mxx, wf + 4.f);
    mny = fmaxf(mny, -4.f); mxy = fminf(mxy, hf + 4.f);
    o.xlo = max((int)floorf(mnx) - 1, 0); o.xhi = min((int)floorf(mxx) + 2, a.w - 1);
    o.ylo = max((int)floorf(mny) - 1, 0); o.yhi = min((int)floorf(mxy) + 2, a.h - 1);
    if (o.xlo > o.xhi || o.ylo > o.yhi) { o = Box{1, 0, 1, 0}; return 0; }
    return 1;
}

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// |s| or s*s accumulated into `acc` (metric fixed at compile time: no branch inside the channel loop)
template <int DIST>
__device__ __forceinline__ float dist_acc(float s, float acc) {
    if constexpr (DIST == NRGBD_DIST_L2) return __builtin_fmaf(s, s, acc);
    else return acc + fabsf(s);
}

template <int CP4, int DIST>
__global__ __launch_bounds__(256, 2) void costvol_lds(const CostvolArgs a) {
    using Cfg = LdsCfg<CP4>;
    constexpr int NCB = Cfg::NCB, NS = Cfg::NS, S4 = Cfg::S4, PMAX = Cfg::PMAX, MAXIT = Cfg::MAXIT;
    extern __shared__ __attribute__((aligned(16))) float4 smem4[];
    int* bb = reinterpret_cast<int*>(smem4 + kPatchF4);  // [kKG][4] = x_lo, x_hi, y_lo, y_hi per candidate

    const int tid = threadIdx.x;
    const int tiles_x = (a.w + kTile - 1) / kTile;
    // Workgroup order.  The hardware deals workgroup ids round-robin to the 8 XCDs; the candidate groups of one tile
    // stage almost the same source texels, so (order bit set, tile count divisible by 8) XCD k owns a contiguous eighth
    // of the TILES and walks it group by group — nearest (most expensive) groups first, as in the plain order, but a
    // tile's groups now run on one XCD within a few dozen workgroups of each other and share its L2.
    int bx = blockIdx.x, by = blockIdx.y;
    if (NRGBD_DBG(a, 4) && (gridDim.x & 7) == 0) {
        const int lin = blockIdx.y * gridDim.x + blockIdx.x;
        const int xcd = lin & 7, u = lin >> 3, tpx = gridDim.x >> 3;   // tiles per XCD
        by = u / tpx;
        bx = xcd * tpx + (u - by * tpx);
    }
    if (NRGBD_DBG(a, 0xff00)) { if (by != ((a.debug >> 8) & 0xff) - 1) return; }   // developer: time one candidate group alone
    const int tx = bx % tiles_x, ty = bx / tiles_x;
    // blockIdx.y -> candidate range.  On grids too small to fill the chip (tiles x D/8 < 4 workgroups per CU)
    // the first kSingles candidates — the nearest planes for an increasing d_candi: large, fast-moving
    // footprints that are staged one by one — get a workgroup each, so that this heavy serial work spreads
    // over idle CUs (config S: 264 -> 110 us).  On large grids the chip is throughput-bound and plain groups
    // of kKG share more staging.  A scheduling choice only: results do not depend on it.
    const int nsingle = a.nsingle;
    const int k0 = (by < nsingle) ? by : nsingle + (by - nsingle) * kKG;
    const int nk = (by < nsingle) ? 1 : min(kKG, a.D - k0);

    // Lane -> pixel map.  ds_read_b128 is serviced in four 16-lane groups {0-3,12-15,20-27},
    // {4-11,16-19,28-31}, {32-35,44-47,52-59}, {36-43,48-51,60-63} (MI355X_MICROARCH.md §LDS); each
    // group is given 16 consecutive pixels of ONE tile row, whose taps are (nearly) 16 consecutive
    // texels of one patch row = 16 distinct bank quads at an odd texel stride.  A wave covers 4 rows.
    const int lane = tid & 63, l5 = lane & 31;
    const int grp = (l5 >= 4 && l5 < 12) || (l5 >= 16 && l5 < 20) || (l5 >= 28);
    const int col = (l5 < 4) ? l5 : (l5 < 12) ? l5 - 4 : (l5 < 16) ? l5 - 8 : (l5 < 20) ? l5 - 8 : (l5 < 28) ? l5 - 12 : l5 - 16;
    const int x = tx * kTile + col, y = ty * kTile + (tid >> 6) * 4 + (lane >> 5) * 2 + grp;
    const bool inside = (x < a.w) && (y < a.h);
    const int xc = min(x, a.w - 1), yc = min(y, a.h - 1);
    const size_t hw = (size_t)a.h * a.w;
    const size_t p = (size_t)yc * a.w + xc;
    const float wf = (float)a.w, hf = (float)a.h;
    const bool align = a.align != 0;

    const float rx = a.rays[p], ry = a.rays[hw + p], rz = a.rays[2 * hw + p];
    const f32x4* refp = reinterpret_cast<const f32x4*>(a.ref + p * a.Cp);
    f32x4 r[CP4];
#pragma unroll
    for (int i = 0; i < CP4; ++i) r[i] = refp[i];
    const int tail = a.C - 4 * (CP4 - 1);  // valid components of the last 16-B word (1..4)

    float dk[kKG], tot[kKG];
#pragma unroll
    for (int j = 0; j < kKG; ++j) {
        dk[j] = a.d_candi[min(k0 + j, a.D - 1)];
        tot[j] = 0.f;
    }

    for (int v = 0; v < a.V; ++v) {
        const float* KRv = a.KR + 9 * v;
        const float* Ktv = a.Kt + 3 * v;
        const float* sv = a.src + (size_t)v * hw * a.Cp;

        // ---- footprint of the whole tile per candidate (threads 0..7) ----
        const int tx0 = tx * kTile, tx1 = min(tx * kTile + kTile - 1, a.w - 1);
        const int ty0 = ty * kTile, ty1 = min(ty * kTile + kTile - 1, a.h - 1);
        if (tid < kKG) {
            Box o{1, 0, 1, 0};
            if (tid < nk) region_box(a, KRv, Ktv, a.d_candi[k0 + tid], tx0, tx1, ty0, ty1, o);
            bb[tid * 4 + 0] = o.xlo; bb[tid * 4 + 1] = o.xhi; bb[tid * 4 + 2] = o.ylo; bb[tid * 4 + 3] = o.yhi;
        }
        __syncthreads();

        const SweepTerm st = make_sweep_term(KRv, Ktv, rx, ry, rz);
        float ixs[kKG], iys[kKG], acc[kKG];
#pragma unroll
        for (int j = 0; j < kKG; ++j) {
            sweep_sample_pos(st, dk[j], a.cx, a.cy, wf, hf, align, ixs[j], iys[j]);
            acc[j] = 0.f;
        }

        int j0 = 0;
        while (j0 < nk) {  // block-uniform
            // largest aligned power-of-two run of candidates whose united footprint fits the patch
            int n = kKG, xlo = 0, xhi = -1, ylo = 0, yhi = -1, area = 0;
            bool fits = false;
            for (; n >= 1; n >>= 1) {
                if ((j0 & (n - 1)) || j0 + n > nk) continue;
                xlo = 1 << 30; ylo = 1 << 30; xhi = -1; yhi = -1;
                for (int j = j0; j < j0 + n; ++j) {
                    const int bx0 = __builtin_amdgcn_readfirstlane(bb[j * 4 + 0]);
                    const int bx1 = __builtin_amdgcn_readfirstlane(bb[j * 4 + 1]);
                    const int by0 = __builtin_amdgcn_readfirstlane(bb[j * 4 + 2]);
                    const int by1 = __builtin_amdgcn_readfirstlane(bb[j * 4 + 3]);
                    if (bx0 <= bx1) {
                        xlo = min(xlo, bx0); xhi = max(xhi, bx1);
                        ylo = min(ylo, by0); yhi = max(yhi, by1);
                    }
                }
                if (xhi < 0) { area = 0; fits = true; break; }  // nothing in view
                const long ar = (long)(xhi - xlo + 1) * (yhi - ylo + 1);
                if (ar <= PMAX) { area = (int)ar; fits = true; break; }
            }
            // A candidate whose whole-tile footprint overflows the patch even alone (zoom > ~1.1: the nearest planes
            // under forward motion) is gathered straight from L2/HBM by all 256 threads; pixels whose four taps are
            // all out of view (most of them on such planes) skip their loads.  Splitting the tile into strips and
            // staging each strip's footprint was tried first: 1.5x slower on those candidates, the threads outside
            // the strip idle through every stage.
            if (!fits) {
#pragma unroll
                for (int j = 0; j < kKG; ++j)
                    if (j == j0) acc[j] += gather_point(sv, a.ref + p * a.Cp, ixs[j], iys[j], a.w, a.h, a.Cp, a.C, a.dist);
                j0 += 1;
                continue;
            }
            const int cols = (area > 0) ? (xhi - xlo + 1) : 1;
            const unsigned magic = (cols > 1) ? (0xFFFFFFFFu / (unsigned)cols + 1u) : 0u;

            static_for<NCB>([&](auto cbc) {
                constexpr int cb = decltype(cbc)::value;
                constexpr int W0 = Cfg::w0(cb);                 // first word staged
                constexpr int L0 = Cfg::first(cb) - W0;         // local index of the first word computed
                constexpr int n4 = Cfg::count(cb);              // words computed
                if (area > 0 && !NRGBD_DBG(a, 1)) {
                    // ---- stage: HBM -> LDS, [texel][NS words]; consecutive lanes take consecutive 16-B
                    // words; every load is issued before the first LDS write (one memory round trip) ----
                    const int total = area * NS;
                    constexpr int BATCH = 16;  // loads in flight per thread (16 x 16 B = 64 VGPRs): one memory round trip per stage
#pragma unroll
                    for (int it0 = 0; it0 < MAXIT; it0 += BATCH) {
                        if ((tid & ~63) + 256 * it0 >= total) continue;  // wave-uniform: nothing left for this wave
                        float4 tmp[BATCH];
#pragma unroll
                        for (int u = 0; u < BATCH; ++u) {
                            // lanes past the end re-load the last word (always in range) so that tmp[] stays
                            // in registers; only the LDS write below is predicated
                            const int i = min(tid + 256 * (it0 + u), total - 1);
                            const int q = i / NS, jw = i - q * NS;
                            const int qy = (cols > 1) ? (int)__umulhi((unsigned)q, magic) : q;
                            const int qx = q - __mul24(qy, cols);
                            // 32-bit element offset from the (uniform) view base: SGPR base + VGPR offset loads
                            const unsigned off = (unsigned)__mul24(__mul24(ylo + qy, a.w) + (xlo + qx), a.Cp) + 4u * (unsigned)(W0 + jw);
                            tmp[u] = *reinterpret_cast<const float4*>(sv + off);
                        }
#pragma unroll
                        for (int u = 0; u < BATCH; ++u) {
                            const int i = tid + 256 * (it0 + u);
                            if (i < total) {
                                const int q = i / NS;
                                smem4[(S4 == NS) ? i : q * S4 + (i - q * NS)] = tmp[u];
                            }
                        }
                    }
                }
                __syncthreads();
#pragma unroll
                for (int j = 0; j < kKG; ++j) {
                    if (j < j0 || j >= j0 + n || NRGBD_DBG(a, 2)) continue;  // uniform
                    float part = 0.f;
                    if (area > 0) {
                        float ix = ixs[j], iy = iys[j];
                        // opaque copies: stop the compiler from keeping 8 candidates x (4 weights + 4
                        // addresses) alive across the channel blocks (that spills); recomputing is ~25 VALU
                        asm volatile("" : "+v"(ix), "+v"(iy));
                        const float x0f = floorf(ix), y0f = floorf(iy);
                        const float fx = ix - x0f, fy = iy - y0f, ex = 1.f - fx, ey = 1.f - fy;
                        const float x1f = x0f + 1.f, y1f = y0f + 1.f;
                        const bool vx0 = (x0f >= 0.f) && (x0f <= wf - 1.f), vx1 = (x1f >= 0.f) && (x1f <= wf - 1.f);
                        const bool vy0 = (y0f >= 0.f) && (y0f <= hf - 1.f), vy1 = (y1f >= 0.f) && (y1f <= hf - 1.f);
                        Bilinear b;
                        b.nw = (vx0 && vy0) ? ey * ex : 0.f; b.ne = (vx1 && vy0) ? ey * fx : 0.f;
                        b.sw = (vx0 && vy1) ? fy * ex : 0.f; b.se = (vx1 && vy1) ? fy * fx : 0.f;
                        // patch-relative texel indices; a tap that is out of view reads texel 0 with weight 0
                        const int xo0 = vx0 ? (int)x0f - xlo : 0, xo1 = vx1 ? (int)x1f - xlo : 0;
                        const int yo0 = vy0 ? __mul24((int)y0f - ylo, cols) : 0, yo1 = vy1 ? __mul24((int)y1f - ylo, cols) : 0;
                        const int last = area - 1;  // belt and braces: never address outside the patch
                        const f32x4* sm = reinterpret_cast<const f32x4*>(smem4);
                        const f32x4* tnw = sm + __mul24(min(max(yo0 + xo0, 0), last), S4);
                        const f32x4* tne = sm + __mul24(min(max(yo0 + xo1, 0), last), S4);
                        const f32x4* tsw = sm + __mul24(min(max(yo1 + xo0, 0), last), S4);
                        const f32x4* tse = sm + __mul24(min(max(yo1 + xo1, 0), last), S4);
                        // Packed fp32 math on the (x,y) and (z,w) halves of each 16-B word (v_pk_fma_f32: two
                        // channels per instruction, adjacent registers, no shuffles); two independent
                        // 2-wide partial sums keep the FMA chains short.
                        const f32x2 wnw = {b.nw, b.nw}, wne = {b.ne, b.ne}, wsw = {b.sw, b.sw}, wse = {b.se, b.se};
                        f32x2 pa = {0.f, 0.f}, pb = {0.f, 0.f};
                        // LDS reads are software-pipelined in groups of GW words: the reads of group g+1
                        // are issued before the math of group g (2 x GW x 4 taps x 16 B = 64 VGPRs in
                        // flight at most — unbounded hoisting of all 36 reads spills)
                        constexpr int GW = 2;
                        constexpr int NG = (n4 + GW - 1) / GW;
                        f32x4 buf[2][GW][4];
                        auto issue = [&](int g, int slot) {
#pragma unroll
                            for (int u = 0; u < GW; ++u) {
                                const int i = g * GW + u;
                                if (i < n4) {
                                    buf[slot][u][0] = tnw[L0 + i]; buf[slot][u][1] = tne[L0 + i];
                                    buf[slot][u][2] = tsw[L0 + i]; buf[slot][u][3] = tse[L0 + i];
                                }
                            }
                        };
                        issue(0, 0);
#pragma unroll
                        for (int g = 0; g < NG; ++g) {
                            if (g + 1 < NG) issue(g + 1, (g + 1) & 1);
                            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                            for (int u = 0; u < GW; ++u) {
                                const int i = g * GW + u;
                                if (i >= n4) continue;
                                const f32x4 A = buf[g & 1][u][0], B = buf[g & 1][u][1], Cc = buf[g & 1][u][2], Dd = buf[g & 1][u][3];
                                const f32x4 rr = r[Cfg::first(cb) + i];
                                f32x2 lo = A.xy * wnw, hi = A.zw * wnw;
                                lo = __builtin_elementwise_fma(B.xy, wne, lo); hi = __builtin_elementwise_fma(B.zw, wne, hi);
                                lo = __builtin_elementwise_fma(Cc.xy, wsw, lo); hi = __builtin_elementwise_fma(Cc.zw, wsw, hi);
                                lo = __builtin_elementwise_fma(Dd.xy, wse, lo); hi = __builtin_elementwise_fma(Dd.zw, wse, hi);
                                lo = lo - rr.xy; hi = hi - rr.zw;
                                if (Cfg::first(cb) + i == CP4 - 1) {  // last word of the texel: only `tail` components count
                                    if (tail < 2) lo.y = 0.f;
                                    if (tail < 3) hi.x = 0.f;
                                    if (tail < 4) hi.y = 0.f;
                                }
                                if constexpr (DIST == NRGBD_DIST_L2) {
                                    pa = __builtin_elementwise_fma(lo, lo, pa);
                                    pb = __builtin_elementwise_fma(hi, hi, pb);
                                } else {
                                    pa = pa + __builtin_elementwise_abs(lo);
                                    pb = pb + __builtin_elementwise_abs(hi);
                                }
                            }
                            __builtin_amdgcn_sched_barrier(0);
                        }
                        pa = pa + pb;
                        part = pa.x + pa.y;
                    } else {  // wholly out of view: every tap is zero, distance to the zero vector
#pragma unroll
                        for (int i = 0; i < n4; ++i) {
                            const f32x4 rr = r[Cfg::first(cb) + i];
                            const int ncomp = (Cfg::first(cb) + i == CP4 - 1) ? tail : 4;
                            const float c4[4] = {rr.x, rr.y, rr.z, rr.w};
#pragma unroll
                            for (int e = 0; e < 4; ++e)
                                if (e < ncomp) {
                                    const float s = 0.f - c4[e];
                                    part = dist_acc<DIST>(s, part);
                                }
                        }
                    }
                    acc[j] += part;
                }
                __syncthreads();
            });
            j0 += n;
        }
#pragma unroll
        for (int j = 0; j < kKG; ++j) tot[j] = tot[j] + acc[j] / a.sigma;  // homography.py:325, views in order
        __syncthreads();  // bb is rewritten by the next view
    }

    float* out = a.out_cost ? a.out_cost : a.out_logp;
    if (inside) {
#pragma unroll
        for (int j = 0; j < kKG; ++j)
            if (j < nk) out[(size_t)(k0 + j) * hw + p] = tot[j];
    }
}

bool costvol_lds_supported(int cp4) {
    switch (cp4) {
        case 1: case 2: case 3: case 4: case 8: case 9: case 16: case 17: return true;
        default: return false;
    }
}

int launch_costvol_lds(const CostvolArgs& args, hipStream_t stream) {
    CostvolArgs a = args;
    const int tiles = ceil_div(a.w, kTile) * ceil_div(a.h, kTile);
    const bool singles = (long)tiles * ceil_div(a.D, kKG) < 4 * 256 || NRGBD_DBG(a, 8);  // under-filled chip
    const int nsingle = singles ? (a.D < kSingles ? a.D : kSingles) : 0;
    a.nsingle = nsingle;
    const dim3 grid(tiles, nsingle + ceil_div(a.D - nsingle, kKG));
    // 64,640 B: below the 64 KiB that needs no opt-in attribute, two workgroups per CU
    const size_t lds = (size_t)kPatchF4 * sizeof(float4) + kKG * 4 * sizeof(int);
#define NRGBD_LDS_CASE(N)                                                                              \
    case N:                                                                                    \
        if (a.dist == NRGBD_DIST_L2)                                                           \
            hipLaunchKernelGGL((costvol_lds<N, NRGBD_DIST_L2>), grid, dim3(256), lds, stream, a); \
        else                                                                                   \
            hipLaunchKernelGGL((costvol_lds<N, NRGBD_DIST_L1>), grid, dim3(256), lds, stream, a); \
        break;
    switch (a.Cp >> 2) {
        NRGBD_LDS_CASE(1)
        NRGBD_LDS_CASE(2)
        NRGBD_LDS_CASE(3)
        NRGBD_LDS_CASE(4)
        NRGBD_LDS_CASE(8)
        NRGBD_LDS_CASE(9)
        NRGBD_LDS_CASE(16)
        NRGBD_LDS_CASE(17)
        default: return NRGBD_E_SHAPE;
    }
#undef NRGBD_LDS_CASE
    NRGBD_CHECK_LAUNCH();
    return NRGBD_OK;
}

}  // namespace nrgbd
