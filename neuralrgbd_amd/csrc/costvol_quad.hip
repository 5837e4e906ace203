// costvol_quad.hip — generation 3 of the fused plane-sweep cost volume: FOUR LANES PER (pixel, candidate).
//
// Why (profiles/r1_pmc_summary.txt): generation 2 gives every lane one reference pixel and keeps its
// 68-channel texel in 68 VGPRs; with the staging batch and the tap pipeline that is 246 VGPRs = 2 waves per SIMD, its
// ds_read_b128 taps conflict on 25 % of the LDS cycles (16 lanes of a service group read 16 unrelated texels), and a
// workgroup alternates stage -> barrier -> math -> barrier with nothing to overlap them.  Here the 64 feature channels of a
// texel are split over the 4 lanes of a quad:
//
//   quad   = one reference pixel (8x8-pixel tile = 64 quads = 256 threads), walking the depth candidates;
//   lane j = feature words {j, 4+j, 8+j, 12+j} of that pixel (16 VGPRs of reference instead of 68), visited in an order
//            rotated by the quad's index: at step s quad i reads quarter (s + i) & 3 of its tap texel, so the four quads that
//            one ds_read_b128 service group holds ({0-3,12-15,20-27}, ... = quads {0,3,5,6}, {1,2,4,7}) always read FOUR
//            DIFFERENT 64-byte quarters => different banks whatever the four texels are: conflict-free BY CONSTRUCTION (LDS
//            texel stride 256 B), not by the luck of the homography being close to a translation;
//   the 4-tap interpolation of a lane's 16 channels needs no data from other lanes; the channel sum of (s - r)^2 is finished
//   with two DPP quad-permute adds (the wavefront-shuffle reduction);
//   the sampling coordinates (2 IEEE divisions each for x and y: the expensive scalar part) are computed ONCE per
//   (pixel, candidate): lane j of the quad does candidate 4g+j, then the record (4 weights + patch address) is broadcast
//   inside the quad with DPP quad_perm [i,i,i,i] while the four lanes evaluate candidate 4g+i together;
//   the RGB word (channels 64..66) of candidate 4g+j is done by lane j alone from a separate 16-B plane of the patch.
//
// Source texels reach the LDS by global_load_lds (no VGPR round trip): a wave instruction drops 4 texels x 256 B in lane
// order, which IS the patch layout.  The patch is the union footprint of a run of 2/4/8 candidates for one view, with a
// one-texel apron that may lie outside the image (filled from the clamped coordinate; its weight is zero), so the four taps
// of a pixel are {A, A+256, A+pitch, A+pitch+256}: one broadcast address per candidate instead of four clamped ones.
// Candidates whose footprint does not fit even as a pair (nearest planes: large, fast-moving footprints) are evaluated
// straight from L1/L2 with the same quad layout — a quad's taps are 64-byte contiguous runs, which the texture path serves at
// 4 lanes/clk (the generation-1 gather paid 1 lane/clk) — so they need neither a patch nor a barrier.
// 165 VGPRs => 3 workgroups (12 waves) per CU; one workgroup owns ALL candidates of its tile (when the grid fills the chip),
// so log_softmax over depth is taken in the same launch (models/basic.py:299-300).
//
// Arithmetic per tap / channel is the same as generations 1-2 (common.hpp helpers, same operation sequence for the
// coordinates); only the order of the channel sum differs (16 channels per lane, then the quad), like any other reduction order.
#include "costvol.hpp"

// developer builds: phase clocks of a workgroup as seen by its thread 0 (a.trace, tools/cv_trace.py); compiled out of the product
#ifdef NRGBD_DEV
#define CVT_DECL long long cvt_[24] = {0}; long long cvt_t0_ = 0; const bool cvt_on_ = a.trace != nullptr && threadIdx.x == 0; \
    if (cvt_on_) { cvt_[0] = wall_clock64(); cvt_[2] = -clock64(); }
#define CVT_BEGIN() do { if (cvt_on_) cvt_t0_ = clock64(); } while (0)
#define CVT_END(slot) do { if (cvt_on_) cvt_[slot] += clock64() - cvt_t0_; } while (0)
#define CVT_COUNT(slot, n) do { if (cvt_on_) cvt_[slot] += (n); } while (0)
#define CVT_FLUSH() do { if (cvt_on_) { cvt_[1] = wall_clock64(); cvt_[2] += clock64(); \
    unsigned hw_; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw_)); cvt_[17] = hw_; \
    unsigned xcc_; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc_)); cvt_[18] = xcc_; \
    for (int q_ = 0; q_ < 24; ++q_) a.trace[(size_t)blockIdx.x * 24 + q_] = cvt_[q_]; } } while (0)
#else
#define CVT_DECL
#define CVT_BEGIN() do {} while (0)
#define CVT_END(slot) do {} while (0)
#define CVT_COUNT(slot, n) do {} while (0)
#define CVT_FLUSH() do {} while (0)
#endif

// A/B build knob (neuralrgbd_amd.build.build_variant): wave priority by progress — 1 (product) = 3 - quarter of the workgroup's own
// (view, candidate) list already done, re-evaluated at every run; 0 = none; 2 = per view only (3 - view, V = 4)
#ifndef NRGBD_CV_PRIO
#define NRGBD_CV_PRIO 1
#endif

namespace nrgbd {

namespace {

constexpr int kQT = 8;            // tile edge
constexpr int kQRun = 32;         // candidates per staged run at most (groups of 4): two 16-lane DPP rows of footprint boxes
// texels of the patch (x 272 B): 156 texels = 42 KB, + 8.4 KB of cost accumulators [64 pixels][32 + 1 candidates], + boxes and candidates = 52.2 KB
// -> 3 workgroups per CU.  Round 5: the views' partial costs of a PASS (32 candidates of the workgroup, all views) meet in these LDS accumulators
// and leave them once; until then a candidate's cost was read-modify-written in global memory once per view (188-texel patch): WRITE_SIZE 53 -> 18.5 MB
// per launch, 236 -> 233.5 us at config B, 50.5 -> 49.1 at S, 226 -> 223 at H (same chip).  (Round 3's generation 4 did the same for all D
// candidates at once next to a 128-texel patch of LOOSE boxes: 415 vs 281 us — it was the patch size, not the accumulators.)
constexpr int kQPatch3 = 156;
constexpr int kQPass = 32;        // candidates per pass (= the longest run)
constexpr int kQAccPitch = kQPass + 1;
constexpr int kQFeatBytes = 256;  // feature plane: 16 words of 16 B per texel

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int N, typename F>
__device__ __forceinline__ void qstatic_for(F&& f) {
    if constexpr (N > 0) {
        qstatic_for<N - 1>(f);
        f(std::integral_constant<int, N - 1>{});
    }
}

template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
template <int CTRL>
__device__ __forceinline__ int dpp_i(int v) {
    return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, true);
}
// quad_perm controls: broadcast lane I of every quad; butterfly partners
template <int I> constexpr int kBcast = I * 0x55;
constexpr int kXor1 = 0xB1;  // [1,0,3,2]
constexpr int kXor2 = 0x4E;  // [2,3,0,1]

struct TapW { float nw, ne, sw, se; };

// weights with zeros padding (a corner outside the image contributes nothing) + the un-clamped integer corner
__device__ __forceinline__ TapW tap_weights(float ix, float iy, float wf, float hf, float& x0f, float& y0f, bool& any) {
    x0f = floorf(ix); y0f = floorf(iy);
    const float fx = ix - x0f, fy = iy - y0f, ex = 1.f - fx, ey = 1.f - fy;
    const float x1f = x0f + 1.f, y1f = y0f + 1.f;
    const bool vx0 = (x0f >= 0.f) && (x0f <= wf - 1.f), vx1 = (x1f >= 0.f) && (x1f <= wf - 1.f);
    const bool vy0 = (y0f >= 0.f) && (y0f <= hf - 1.f), vy1 = (y1f >= 0.f) && (y1f <= hf - 1.f);
    any = (vx0 || vx1) && (vy0 || vy1);          // some tap lies inside the image (lane masks: scalar-unit work)
    TapW t;
    t.nw = (vx0 && vy0) ? ey * ex : 0.f; t.ne = (vx1 && vy0) ? ey * fx : 0.f;
    t.sw = (vx0 && vy1) ? fy * ex : 0.f; t.se = (vx1 && vy1) ? fy * fx : 0.f;
    return t;
}

// one 16-byte word of the 4 taps against the reference word: accumulate dist(sample - ref) of its 4 channels
template <int DIST>
__device__ __forceinline__ void word_acc(const f32x4 A, const f32x4 B, const f32x4 C, const f32x4 D, const f32x4 rr,
                                         const f32x2 wnw, const f32x2 wne, const f32x2 wsw, const f32x2 wse, f32x2& pa,
                                         f32x2& pb) {
    // sample - ref with the reference word as the initial value of the tap chain (one rounding order among many; saves the
    // separate subtraction: 10 instead of 12 packed operations per word)
    f32x2 lo = __builtin_elementwise_fma(A.xy, wnw, -rr.xy), hi = __builtin_elementwise_fma(A.zw, wnw, -rr.zw);
    lo = __builtin_elementwise_fma(B.xy, wne, lo); hi = __builtin_elementwise_fma(B.zw, wne, hi);
    lo = __builtin_elementwise_fma(C.xy, wsw, lo); hi = __builtin_elementwise_fma(C.zw, wsw, hi);
    lo = __builtin_elementwise_fma(D.xy, wse, lo); hi = __builtin_elementwise_fma(D.zw, wse, hi);
    if constexpr (DIST == NRGBD_DIST_L2) {
        pa = __builtin_elementwise_fma(lo, lo, pa);
        pb = __builtin_elementwise_fma(hi, hi, pb);
    } else {
        pa = pa + __builtin_elementwise_abs(lo);
        pb = pb + __builtin_elementwise_abs(hi);
    }
}

// the RGB word (channels 64.. : `tail` of them count) of one candidate, all four taps by one lane
template <int DIST>
__device__ __forceinline__ float rgb_word(const f32x4 A, const f32x4 B, const f32x4 C, const f32x4 D, const f32x4 rr,
                                          const TapW t, int tail) {
    float s[4];
    s[0] = __builtin_fmaf(D.x, t.se, __builtin_fmaf(C.x, t.sw, __builtin_fmaf(B.x, t.ne, A.x * t.nw))) - rr.x;
    s[1] = __builtin_fmaf(D.y, t.se, __builtin_fmaf(C.y, t.sw, __builtin_fmaf(B.y, t.ne, A.y * t.nw))) - rr.y;
    s[2] = __builtin_fmaf(D.z, t.se, __builtin_fmaf(C.z, t.sw, __builtin_fmaf(B.z, t.ne, A.z * t.nw))) - rr.z;
    s[3] = __builtin_fmaf(D.w, t.se, __builtin_fmaf(C.w, t.sw, __builtin_fmaf(B.w, t.ne, A.w * t.nw))) - rr.w;
    float acc = 0.f;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float term = (DIST == NRGBD_DIST_L2) ? s[e] * s[e] : fabsf(s[e]);
        acc = acc + ((e < tail) ? term : 0.f);
    }
    return acc;
}

}  // namespace

// TAIL: valid channels of the texel's 17th word (Cp = 68: channels 64..66 = pooled RGB => 3; -1 = a.C - 64 at run time);
// 0 = no 17th word (Cp = 64).  ALIGN: grid_sample's align_corners.
// A candidate's cost is accumulated over the views in LDS (ldsA) by the lane that owns it in each view and written once per pass.
template <int DIST, int TAIL, bool ALIGN, int PATCH, int WGS>
__global__ __launch_bounds__(256, WGS) void costvol_quad(const CostvolArgs a) {
    constexpr bool EXTRA = TAIL != 0;
    constexpr int kQPatch = PATCH, kQPatchR = (PATCH + 63) / 64 * 64;   // RGB plane: a wave instruction fills 64 slots
    extern __shared__ __attribute__((aligned(16))) char smem[];
    CVT_DECL
    char* ldsF = smem;                                                  // [kQPatch][256 B]
    char* ldsR = smem + kQPatch * kQFeatBytes;                          // [kQPatchR][16 B]
    int4* boxes = reinterpret_cast<int4*>(ldsR + kQPatchR * 16);        // [kQPass] footprint of the tile per candidate of the pass (this view)
    float* dcand = reinterpret_cast<float*>(boxes + kQPass);            // [D] depth candidates
    float* ldsA = dcand + ((a.D + 3) & ~3);                             // [64 pixels][kQAccPitch] partial costs of the pass, summed over the views
    float* red = reinterpret_cast<float*>(smem);                        // softmax scratch (the patch is dead by then)

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int quad = tid >> 2, j = tid & 3, i3 = quad & 3;
    // workgroup -> (tile, candidate chunk); XCD k (= blockIdx % 8) owns a contiguous eighth of the tile list so that the
    // source rows a band of tiles samples stay in ONE 4 MB L2
    // workgroup -> (tile, candidate chunk).  XCD k (= blockIdx % 8) owns a contiguous eighth of the TILE list so that the source rows
    // a band of tiles samples stay in ONE 4 MB L2; with several chunks per tile the XCD's workgroups are ordered chunk-major —
    // every tile's NEAREST chunk first: those are the expensive ones (short runs, unstaged groups: 2-3x the time of a far chunk,
    // profiles/r5_costvol_diag_before.txt), and the hardware hands out workgroups in index order as slots free up, so the
    // cheap far chunks fill in behind the slow near ones (longest-processing-time-first) instead of a slow chunk starting last.
    const int ntiles = gridDim.x / a.nchunk;
    int tile = blockIdx.x / a.nchunk, chunk = blockIdx.x - tile * a.nchunk;
    if ((ntiles & 7) == 0) {
        const int per = ntiles >> 3;             // tiles of one XCD = its band of the tile list
        const int uu = blockIdx.x >> 3;
        chunk = uu / per;
        int u = uu - chunk * per;
        // Inside the band the tiles are handed out in a scrambled order (developer bit 64 turns it off): the workgroups that
        // share a CU (consecutive or 32 apart in an XCD's dispatch order) then come from tiles ~5 columns / a row + 11
        // columns apart instead of neighbours.  A tile's cost (how many near-plane runs go to global memory) varies
        // smoothly over the image, so neighbours on one CU could make slow CUs and fast CUs.  Measured: 277 vs 280 us at
        // config B (within run-to-run noise) — per-CU load imbalance is not what limits the kernel.
        // (Round 5 tried giving every XCD two half-bands, one from the front of the list and its mirror image from the back, so
        // that a linear cost gradient over the image cancels: 236 vs 234 us, no effect — the workgroups' durations spread 1.9x
        // inside every XCD, profiles/r5_costvol_limits.txt.)
        if ((per & 31) == 0 && !NRGBD_DBG(a, 64)) {
            const int r = u >> 5, c = u & 31;
            u = (r << 5) | ((c * 5 + r * 11) & 31);
        }
        tile = (blockIdx.x & 7) * per + u;
    } else if ((gridDim.x & 7) == 0) {
        // a tile count that does not split into eight bands (config H: 300 tiles x 4 chunks): the XCD owns a contiguous eighth of
        // the (tile, chunk) list instead, tile-major — its chunks of one tile sample the same source rows from ONE L2 (without
        // this the four chunks of a tile ran on four XCDs: FETCH_SIZE 50 -> 105 MB per launch at config H)
        const int per = (int)gridDim.x >> 3;
        const int id = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
        tile = id / a.nchunk;
        chunk = id - tile * a.nchunk;
    }
    // Developer switch (NRGBD_ABLATE bit 16): every other workgroup walks its candidates far -> near, to de-phase the
    // staging-bound (near planes) and math-bound (far planes) parts of co-resident workgroups.  Measured: 288 us with,
    // 277 us without at config B — the workgroups are not phase-locked; kept off.
    const int uq = blockIdx.x >> 3;
    const bool rev = NRGBD_DBG(a, 16) && (((uq ^ (uq >> 5)) & 1) != 0);
    const int tiles_x = (a.w + kQT - 1) / kQT;
    const int tx = tile % tiles_x, ty = tile / tiles_x;
    const int kb_all = chunk * a.kchunk, ke_all = min(a.D, kb_all + a.kchunk);

    const int x = tx * kQT + (quad & 7), y = ty * kQT + (quad >> 3);
    const bool inside = (x < a.w) && (y < a.h);
    const int xc = min(x, a.w - 1), yc = min(y, a.h - 1);
    const size_t hw = (size_t)a.h * a.w;
    const size_t p = (size_t)yc * a.w + xc;
    const float wf = (float)a.w, hf = (float)a.h;
    const int tail = TAIL >= 0 ? TAIL : a.C - 64;   // valid channels of the RGB word
    for (int k = tid; k < a.D; k += 256) dcand[k] = a.d_candi[k];

    // per-lane word order: step s -> byte offset of word ((s + i3) & 3) * 4 + j inside a texel's feature plane
    int cofs[4];
    f32x4 rr[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        cofs[s] = (((s + i3) & 3) * 4 + j) * 16;
        rr[s] = *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(a.ref + p * a.Cp) + cofs[s]);
    }
    f32x4 rgbref = {0.f, 0.f, 0.f, 0.f};
    if constexpr (EXTRA) rgbref = *reinterpret_cast<const f32x4*>(a.ref + p * a.Cp + 64);
    const float rx = a.rays[p], ry = a.rays[hw + p], rz = a.rays[2 * hw + p];

    const int tx0 = tx * kQT, tx1 = min(tx * kQT + kQT - 1, a.w - 1);
    const int ty0 = ty * kQT, ty1 = min(ty * kQT + kQT - 1, a.h - 1);
    float* out = a.out_cost ? a.out_cost : a.out_logp;

    // ---- evaluation of one group of NI <= 4 candidates (lane j owns candidate k0 + j) for one view --------------------------
    // STAGED: taps from the LDS patch (box xlo..: pitch = cols); otherwise straight from global memory.  The NI x 4
    // (candidate, step) tap fetches run as one software pipeline, PD steps ahead of the math (LDS: 1, global memory: 2).
    auto group = [&](auto staged_c, auto ni_c, const float* sv, const SweepTerm& st, int k0, int ncand, int xlo, int xhi,
                     int ylo, int yhi, int cols, bool& escaped) -> float {
        constexpr bool STAGED = decltype(staged_c)::value;
        constexpr int NI = decltype(ni_c)::value;
        constexpr int PD = STAGED ? 1 : 2;
        const int kc = min(k0 + j, k0 + ncand - 1);          // lanes beyond the group repeat its last candidate
        float ix, iy, x0f, y0f;
        sweep_sample_pos_fast<ALIGN>(st, dcand[kc], a.cx, a.cy, a.rcx, a.rcy, wf, hf, ix, iy);
        bool any_tap;
        const TapW tw = tap_weights(ix, iy, wf, hf, x0f, y0f, any_tap);
        // tap addresses: STAGED one patch address (apron => the 3 other taps are +256, +pitch, +pitch+256);
        // global: four clamped texel offsets
        int adr[4];
        if constexpr (STAGED) {
            const float xcl = fminf(fmaxf(x0f, (float)xlo), (float)(xhi - 1));           // NaN -> xlo
            const float ycl = fminf(fmaxf(y0f, (float)ylo), (float)(yhi - 1));
            const int xi = (int)xcl - xlo, yi = (int)ycl - ylo;
            adr[0] = __mul24(yi, cols) + xi;                  // texel index in the patch
            // A tap with a non-zero weight that the clamp moved has LEFT the box the tile's corners predicted.  The box is exact for
            // the pinhole ray table of the reference (warping/View.py:16-62: rays affine in (x, y), z = 1 — a homography maps the convex
            // tile into the convex hull of its corners' images) and carries 1/64 texel of guard; another table (unit-norm or
            // distortion-corrected rays: the C ABI takes any) can bend a tile's image out of it.  Legitimate clamps — a box clipped to
            // the image's one-texel apron — only ever move taps whose weights are all zero.  The caller re-evaluates the group from
            // global memory when any lane of the wave reports an escape: 2 compares per (pixel, candidate).
            escaped = any_tap && ((xcl != x0f) || (ycl != y0f));
            adr[1] = adr[2] = adr[3] = 0;
        } else {
            const int xa = (int)fminf(fmaxf(x0f, 0.f), wf - 1.f), xb = (int)fminf(fmaxf(x0f + 1.f, 0.f), wf - 1.f);
            const int ya = (int)fminf(fmaxf(y0f, 0.f), hf - 1.f), yb = (int)fminf(fmaxf(y0f + 1.f, 0.f), hf - 1.f);
            const int cpb = a.Cp * 4;
            const int ra = __mul24(ya, a.w), rb = __mul24(yb, a.w);          // 24-bit multiplies (full rate): h * w < 2^24, checked by the launcher
            adr[0] = __mul24(ra + xa, cpb); adr[1] = __mul24(ra + xb, cpb);
            adr[2] = __mul24(rb + xa, cpb); adr[3] = __mul24(rb + xb, cpb);
        }
        const int pitchB = cols * kQFeatBytes;
        const char* gsv = reinterpret_cast<const char*>(sv);

        // broadcast tap addresses of the NI candidates (item I = the candidate lane I of the quad prepared)
        int ib[NI][STAGED ? 1 : 4];
        qstatic_for<NI>([&](auto ic) {
            constexpr int I = decltype(ic)::value;
            if constexpr (STAGED) {
                ib[I][0] = dpp_i<kBcast<I>>(adr[0]) * kQFeatBytes;
            } else {
#pragma unroll
                for (int t = 0; t < 4; ++t) ib[I][t] = dpp_i<kBcast<I>>(adr[t]);
            }
        });
        f32x4 buf[PD + 1][4];
        auto issue = [&](auto sc) {
            constexpr int S = decltype(sc)::value;
            constexpr int I = S >> 2, s4 = S & 3, slot = S % (PD + 1);
            if constexpr (STAGED) {
                const char* t0 = ldsF + ib[I][0] + cofs[s4];
                const char* t1 = t0 + pitchB;
                buf[slot][0] = *reinterpret_cast<const f32x4*>(t0);
                buf[slot][1] = *reinterpret_cast<const f32x4*>(t0 + kQFeatBytes);
                buf[slot][2] = *reinterpret_cast<const f32x4*>(t1);
                buf[slot][3] = *reinterpret_cast<const f32x4*>(t1 + kQFeatBytes);
            } else {
#pragma unroll
                for (int t = 0; t < 4; ++t)
                    buf[slot][t] = *reinterpret_cast<const f32x4*>(gsv + (unsigned)(ib[I][t] + cofs[s4]));
            }
        };
        qstatic_for<(PD < NI * 4 ? PD : NI * 4)>([&](auto sc) { issue(sc); });

        // RGB word of this lane's own candidate (its four taps by this lane alone), overlapped with the first fetches
        float rgbpart = 0.f;
        if constexpr (EXTRA) {
            f32x4 A, B, C, Dd;
            if constexpr (STAGED) {
                const char* r0 = ldsR + adr[0] * 16;
                const char* r1 = r0 + cols * 16;
                A = *reinterpret_cast<const f32x4*>(r0); B = *reinterpret_cast<const f32x4*>(r0 + 16);
                C = *reinterpret_cast<const f32x4*>(r1); Dd = *reinterpret_cast<const f32x4*>(r1 + 16);
            } else {
                const char* g = gsv + kQFeatBytes;
                A = *reinterpret_cast<const f32x4*>(g + (unsigned)adr[0]); B = *reinterpret_cast<const f32x4*>(g + (unsigned)adr[1]);
                C = *reinterpret_cast<const f32x4*>(g + (unsigned)adr[2]); Dd = *reinterpret_cast<const f32x4*>(g + (unsigned)adr[3]);
            }
            rgbpart = rgb_word<DIST>(A, B, C, Dd, rgbref, tw, tail);
        }

        float keep = 0.f;
        f32x2 wnw, wne, wsw, wse, pa, pb;
        qstatic_for<NI * 4>([&](auto sc) {
            constexpr int S = decltype(sc)::value;
            constexpr int I = S >> 2, s4 = S & 3, slot = S % (PD + 1);
            if constexpr (S + PD < NI * 4) issue(std::integral_constant<int, S + PD>{});
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (s4 == 0) {
                const float bnw = dpp_f<kBcast<I>>(tw.nw), bne = dpp_f<kBcast<I>>(tw.ne);
                const float bsw = dpp_f<kBcast<I>>(tw.sw), bse = dpp_f<kBcast<I>>(tw.se);
                wnw = f32x2{bnw, bnw}; wne = f32x2{bne, bne}; wsw = f32x2{bsw, bsw}; wse = f32x2{bse, bse};
                pa = f32x2{0.f, 0.f}; pb = f32x2{0.f, 0.f};
            }
            word_acc<DIST>(buf[slot][0], buf[slot][1], buf[slot][2], buf[slot][3], rr[s4], wnw, wne, wsw, wse, pa, pb);
            if constexpr (s4 == 3) {
                const f32x2 pc = pa + pb;
                float part = pc.x + pc.y;
                part = part + dpp_f<kXor1>(part);              // quad reduction: all four lanes end with the channel sum
                part = part + dpp_f<kXor2>(part);
                keep = (j == I) ? part : keep;
            }
            __builtin_amdgcn_sched_barrier(0);
        });
        return keep + rgbpart;
    };
    using T_ = std::true_type; using F_ = std::false_type;
    using N1 = std::integral_constant<int, 1>; using N2 = std::integral_constant<int, 2>; using N3 = std::integral_constant<int, 3>;
    using N4 = std::integral_constant<int, 4>;

    // Loop order: views OUTER.  All workgroups of an XCD (one band of tiles) sweep the same source view at about the same
    // time, so the band's footprint in ONE view (~2 MB) is what has to live in the 4 MB L2 — with the views inside the
    // candidate loop it would be all V of them (~9 MB) and every patch fill would go to the fabric.  The price is that a
    // candidate's cost has to be accumulated across views: in the LDS accumulators of the PASS (kQPass candidates of this workgroup, all
    // their views), which leave for global memory once at the end of the pass (until round 5: read-modify-written in global memory per view).
    int prio_q = -1;
    // ---- passes of kQPass candidates (the body below is the view loop of ONE pass; it keeps its indentation) ----
    for (int kb = kb_all; kb < ke_all; kb += kQPass) {
    const int ke = min(ke_all, kb + kQPass);
    const int bo = kb;                                  // boxes[] is indexed relative to the pass
    for (int v = 0; v < a.V; ++v) {
#if NRGBD_CV_PRIO == 2
        if (v == 0) __builtin_amdgcn_s_setprio(3); else if (v == 1) __builtin_amdgcn_s_setprio(2); else if (v == 2) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0);
#endif
        const float* KRv = a.KR + 9 * v;
        const float* Ktv = a.Kt + 3 * v;
        const float* sv = a.src + (size_t)v * hw * a.Cp;
        const SweepTerm st = make_sweep_term(KRv, Ktv, rx, ry, rz);
        // ---- footprint of the tile on every candidate plane of this view: thread = (candidate, tile corner); the sampling
        // positions of the 4 corner pixels bound the taps of the whole tile (a homography maps the convex tile onto a convex
        // quadrilateral as long as the plane stays in front of the source camera: `bad` otherwise).  Box = the texels their
        // bounding box touches, inside [-1, w] x [-1, h] (apron), at least 2 x 2 ----
        CVT_BEGIN();
        __syncthreads();                                       // the previous view's boxes / the prologue's dcand
        for (int c0 = kb; c0 < ke; c0 += 64) {
            const int c = c0 + (tid >> 2), cr = tid & 3;
            float fxlo = 0.f, fxhi = 0.f, fylo = 0.f, fyhi = 0.f;
            int bad = 0;
            if (c < ke) {
                const int pxc = (cr & 1) ? tx1 : tx0, pyc = (cr & 2) ? ty1 : ty0;
                const size_t pc = (size_t)pyc * a.w + pxc;
                const SweepTerm sc = make_sweep_term(KRv, Ktv, a.rays[pc], a.rays[hw + pc], a.rays[2 * hw + pc]);
                const float dc = dcand[c];
                const float den = (sc.t1z + sc.t2z * dc) + 1e-10f;
                float ix, iy;
                sweep_sample_pos_fast<ALIGN>(sc, dc, a.cx, a.cy, a.rcx, a.rcy, wf, hf, ix, iy);
                bad = ((den > 0.f) && (fabsf(ix) < 1e8f) && (fabsf(iy) < 1e8f)) ? 0 : 1;
                fxlo = fxhi = ix; fylo = fyhi = iy;
            }
            fxlo = fminf(fxlo, dpp_f<kXor1>(fxlo)); fxhi = fmaxf(fxhi, dpp_f<kXor1>(fxhi));
            fylo = fminf(fylo, dpp_f<kXor1>(fylo)); fyhi = fmaxf(fyhi, dpp_f<kXor1>(fyhi));
            bad |= dpp_i<kXor1>(bad);
            fxlo = fminf(fxlo, dpp_f<kXor2>(fxlo)); fxhi = fmaxf(fxhi, dpp_f<kXor2>(fxhi));
            fylo = fminf(fylo, dpp_f<kXor2>(fylo)); fyhi = fmaxf(fyhi, dpp_f<kXor2>(fyhi));
            bad |= dpp_i<kXor2>(bad);
            if (c < ke && cr == 0) {
                int4 b;
                // TIGHT box (round 5): a pixel's taps are texels floor(i) and floor(i) + 1; over the tile floor(i) ranges over
                // [floor(min), floor(max)] of the corner positions (convexity), so the box is [floor(min), floor(max) + 1] — with
                // 1/64 texel of guard for the fp32 evaluation of interior pixels against the corners' (their positions agree
                // with exact arithmetic to ~1e-4 texel).  Rounds 2-4 carried one more texel on every side: an 8x8 tile at unit
                // scale filled 11 x 11 = 121 texels per candidate instead of 9 x 9 = 81, and a patch held 2/3 of the candidates.
                const float mnx = fmaxf(fxlo, -4.f) - 0.015625f, mxx = fminf(fxhi, wf + 4.f) + 0.015625f;
                const float mny = fmaxf(fylo, -4.f) - 0.015625f, mxy = fminf(fyhi, hf + 4.f) + 0.015625f;
                b.x = min(max((int)floorf(mnx), -1), a.w - 1);
                b.y = min(max((int)floorf(mxx) + 1, b.x + 1), a.w);
                b.z = min(max((int)floorf(mny), -1), a.h - 1);
                b.w = min(max((int)floorf(mxy) + 1, b.z + 1), a.h);
                if (bad) { b.x = -(1 << 20); b.y = 1 << 20; b.z = -(1 << 20); b.w = 1 << 20; }   // unbounded: never fits
                boxes[c - bo] = b;
            }
        }
        __syncthreads();
        CVT_END(3);
        for (int lo = kb, hi = ke; lo < hi;) {
#if NRGBD_CV_PRIO == 1
            {
                // WAVE PRIORITY BY PROGRESS (round 5).  The launch is one round of 3 workgroups per CU (768 tiles = 256 x 3 at config B) and a
                // SIMD arbitrates its three waves by age: the workgroup dispatched first wins every contended issue slot, so the three
                // finished at 350 / 420 / 480 k clocks whatever their tiles (the per-tile cost map has a period of exactly one dispatch round,
                // profiles/r5_costvol_limits.txt) and the last one ran its final quarter alone on the CU, with nothing to hide its LDS latency
                // behind.  s_setprio beats age: a workgroup's priority is 3 minus the quarter of its own (view, candidate) list it has
                // finished, so whoever is ahead yields and the three stay within a quarter of each other.  230 -> 211 us.
                const int total = a.V * (ke_all - kb_all), done = a.V * (kb - kb_all) + v * (ke - kb) + ((rev ? ke - hi : lo - kb));
                const int q = (4 * done) / total;
                if (q != prio_q) {
                    prio_q = q;
                    if (q <= 0) __builtin_amdgcn_s_setprio(3); else if (q == 1) __builtin_amdgcn_s_setprio(2);
                    else if (q == 2) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0);
                }
            }
#endif
            CVT_BEGIN();
            // ---- longest run of the next candidates (up to 32) whose united footprint fits the patch: lane l < 32 of every wave
            // holds the l-th next candidate, inclusive prefix union along the two 16-lane DPP rows (the second row joins the
            // first row's total), every lane tests ITS prefix, and the run length is the number of leading lanes that fit —
            // any length, not only 16 / 8 / 4 / 2 (round 5: one ballot instead of five rounds of four v_readlane + scalar
            // compares; the selection took 8 % of a workgroup's time, profiles/r5_costvol_diag_before.txt) ----
            const int nmax = min(kQRun, hi - lo);
            int4 bx = make_int4(1 << 30, -(1 << 30), 1 << 30, -(1 << 30));
            if (lane < nmax) bx = boxes[(rev ? hi - 1 - lane : lo + lane) - bo];
#define NRGBD_ROW_SHR_UNION(SH)                                                                                   \
            bx.x = min(bx.x, __builtin_amdgcn_update_dpp(1 << 30, bx.x, 0x110 + SH, 0xf, 0xf, false));              \
            bx.y = max(bx.y, __builtin_amdgcn_update_dpp(-(1 << 30), bx.y, 0x110 + SH, 0xf, 0xf, false));           \
            bx.z = min(bx.z, __builtin_amdgcn_update_dpp(1 << 30, bx.z, 0x110 + SH, 0xf, 0xf, false));              \
            bx.w = max(bx.w, __builtin_amdgcn_update_dpp(-(1 << 30), bx.w, 0x110 + SH, 0xf, 0xf, false));
            NRGBD_ROW_SHR_UNION(1) NRGBD_ROW_SHR_UNION(2) NRGBD_ROW_SHR_UNION(4) NRGBD_ROW_SHR_UNION(8)
#undef NRGBD_ROW_SHR_UNION
            if (nmax > 16) {                                   // block-uniform
                const int r0x = __builtin_amdgcn_readlane(bx.x, 15), r0y = __builtin_amdgcn_readlane(bx.y, 15);
                const int r0z = __builtin_amdgcn_readlane(bx.z, 15), r0w = __builtin_amdgcn_readlane(bx.w, 15);
                if (lane >= 16) { bx.x = min(bx.x, r0x); bx.y = max(bx.y, r0y); bx.z = min(bx.z, r0z); bx.w = max(bx.w, r0w); }
            }
            const int bwd = bx.y - bx.x + 1, bht = bx.w - bx.z + 1;      // 2^21 for an unbounded box: test the sides before the product
            const bool fit = lane < nmax && bwd <= kQPatch && bht <= kQPatch && bwd * bht <= kQPatch;
            // prefix unions only grow, so the lanes that fit are a prefix of the wave: its length is the run
            int n = (int)__builtin_ctzll(~__ballot(fit));
            int xlo = 0, xhi = 1, ylo = 0, yhi = 1;
            if (n >= 1) {
                xlo = __builtin_amdgcn_readlane(bx.x, n - 1); xhi = __builtin_amdgcn_readlane(bx.y, n - 1);
                ylo = __builtin_amdgcn_readlane(bx.z, n - 1); yhi = __builtin_amdgcn_readlane(bx.w, n - 1);
            }
            // a single candidate whose footprint fits could be staged as well (developer bit 32): measured 300 us vs 287 us
            // at config B — a patch fill + two barriers for 64 (pixel, candidate) pairs costs more than their 16 global loads
            const bool staged = n >= (NRGBD_DBG(a, 32) ? 1 : 2) && !NRGBD_DBG(a, 1);
            if (!staged) n = min(4, nmax);                     // one group straight from global memory
            const int j0 = rev ? hi - n : lo;                  // the run: candidates j0 .. j0 + n - 1
            const int ngroups = (n + 3) >> 2;
            const int cols = xhi - xlo + 1;
            CVT_END(4);
            if (staged) { CVT_COUNT(n >= 16 ? 10 : n >= 8 ? 11 : n >= 4 ? 12 : 13, 1); CVT_COUNT(15, n); CVT_COUNT(19, (long long)cols * (yhi - ylo + 1)); } else CVT_COUNT(14, 1);

            if (staged) {
                const int area = cols * (yhi - ylo + 1);
                const float inv_cols = 1.0f / (float)cols;     // q / cols = (int)((q + 0.5) * inv_cols): exact for q, cols <= 192
                const int cpb = a.Cp * 4;
                const char* svb = reinterpret_cast<const char*>(sv);
                CVT_BEGIN();
                __syncthreads();                               // the previous patch has been read by every wave
                CVT_END(16);
                CVT_BEGIN();
                // ---- stage: L2 -> LDS without touching VGPRs; a wave instruction = 4 texels x 256 B (feature plane) or
                // 64 texels x 16 B (RGB plane), landing in lane order = patch order ----
                // a patch wholly inside the image (the common case) needs no coordinate clamps: texel (qx, qy) of the patch
                // is at base + qy * row pitch + qx * texel pitch
                const bool interior = xlo >= 0 && ylo >= 0 && xhi <= a.w - 1 && yhi <= a.h - 1;   // block-uniform
                const int rowb = a.w * cpb;
                const unsigned base = (unsigned)((ylo * a.w + xlo) * cpb);
                if (interior) {
                    for (int it = wave; it * 4 < area; it += 4) {
                        const int q = min(it * 4 + (lane >> 4), area - 1);
                        const int qy = (int)(((float)q + 0.5f) * inv_cols), qx = q - __mul24(qy, cols);
                        const unsigned off = base + (unsigned)(__mul24(qy, rowb) + __mul24(qx, cpb) + (lane & 15) * 16);   // 24-bit multiplies: full rate (a row pitch < 16 MB)
                        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(svb + off),
                                                         (__attribute__((address_space(3))) void*)(ldsF + it * 4 * kQFeatBytes),
                                                         16, 0, 0);
                    }
                    if constexpr (EXTRA) {
                        for (int it = wave; it * 64 < area; it += 4) {
                            const int q = min(it * 64 + lane, area - 1);
                            const int qy = (int)(((float)q + 0.5f) * inv_cols), qx = q - __mul24(qy, cols);
                            const unsigned off = base + (unsigned)(__mul24(qy, rowb) + __mul24(qx, cpb) + kQFeatBytes);
                            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(svb + off),
                                                             (__attribute__((address_space(3))) void*)(ldsR + it * 64 * 16),
                                                             16, 0, 0);
                        }
                    }
                } else {
                    for (int it = wave; it * 4 < area; it += 4) {
                        const int q = min(it * 4 + (lane >> 4), area - 1);
                        const int qy = (int)(((float)q + 0.5f) * inv_cols), qx = q - __mul24(qy, cols);
                        const int gx = min(max(xlo + qx, 0), a.w - 1), gy = min(max(ylo + qy, 0), a.h - 1);
                        const unsigned off = (unsigned)(__mul24(gy, rowb) + __mul24(gx, cpb) + (lane & 15) * 16);
                        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(svb + off),
                                                         (__attribute__((address_space(3))) void*)(ldsF + it * 4 * kQFeatBytes),
                                                         16, 0, 0);
                    }
                    if constexpr (EXTRA) {
                        for (int it = wave; it * 64 < area; it += 4) {
                            const int q = min(it * 64 + lane, area - 1);
                            const int qy = (int)(((float)q + 0.5f) * inv_cols), qx = q - __mul24(qy, cols);
                            const int gx = min(max(xlo + qx, 0), a.w - 1), gy = min(max(ylo + qy, 0), a.h - 1);
                            const unsigned off = (unsigned)(__mul24(gy, rowb) + __mul24(gx, cpb) + kQFeatBytes);
                            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(svb + off),
                                                             (__attribute__((address_space(3))) void*)(ldsR + it * 64 * 16),
                                                             16, 0, 0);
                        }
                    }
                }
                CVT_END(5);
                CVT_BEGIN();
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                CVT_END(6);
            }
            CVT_BEGIN();
            if (!NRGBD_DBG(a, 2)) {
#pragma unroll 1
                for (int g = 0; g < ngroups; ++g) {
                    const int k0 = j0 + 4 * g, nc = min(4, n - 4 * g);   // candidates j0 .. j0 + n - 1 = this run
                    float* o = ldsA + quad * kQAccPitch + (min(k0 + j, ke - 1) - kb);   // this lane's candidate of the group; a quad is inside one wave: in-order LDS
                    const float prev = (v > 0) ? *o : 0.f;
                    float acc = 0.f;
                    bool esc = false, from_lds = staged;
#pragma unroll 1
                    for (int attempt = 0; attempt < 2; ++attempt) {
                        if (!from_lds) { acc = group(F_{}, N4{}, sv, st, k0, nc, 0, 0, 0, 0, 0, esc); break; }
                        if (nc == 1) acc = group(T_{}, N1{}, sv, st, k0, 1, xlo, xhi, ylo, yhi, cols, esc);
                        else if (nc == 2) acc = group(T_{}, N2{}, sv, st, k0, 2, xlo, xhi, ylo, yhi, cols, esc);
                        else if (nc == 3) acc = group(T_{}, N3{}, sv, st, k0, 3, xlo, xhi, ylo, yhi, cols, esc);
                        else acc = group(T_{}, N4{}, sv, st, k0, 4, xlo, xhi, ylo, yhi, cols, esc);
                        if (__ballot(esc) == 0) break;          // wave-uniform: the 16 quads of a wave redo the group together
                        CVT_COUNT(20, 1);
                        from_lds = false;                       // a tap escaped the footprint box: this group straight from global memory
                    }
                    if (j < nc) *o = prev + div_by_const(acc, a.sigma, a.rsigma);               // homography.py:325 (/ sigma), views in order
                }
            }
            if (staged) CVT_END(7); else CVT_END(8);
            if (rev) hi -= n; else lo += n;
        }
    }
    // ---- the pass is complete: its costs leave the LDS once (no read-modify-write per view: -3 reads and -3 writes of D x hw floats) ----
    __syncthreads();
    for (int idx = tid; idx < 64 * (ke - kb); idx += 256) {
        const int kk = idx >> 6, q = idx & 63;
        const int xq = tx * kQT + (q & 7), yq = ty * kQT + (q >> 3);
        if (xq < a.w && yq < a.h) out[(size_t)(kb + kk) * hw + (size_t)yq * a.w + xq] = ldsA[q * kQAccPitch + kk];
    }
    __syncthreads();                                    // the accumulators are free for the next pass
    }

    if (!a.fuse_softmax) { CVT_FLUSH(); return; }
    CVT_BEGIN();
    // ---- log_softmax(-cost) over the D candidates of the tile's pixels (models/basic.py:299-300) ----
    // thread = (pixel, quarter of the candidates); the costs were written by other lanes of THIS workgroup: make the stores
    // visible (L2) and read them past the L1
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __syncthreads();
    // thread = (pixel, quarter of the candidates): KMAX candidates per thread — 16 for D <= 64 (the path's configs S / B / K; the D <= 128
    // instantiation spent half of its unrolled iterations on candidates that do not exist), 32 for D <= 128 (config H when unchunked)
    auto softmax = [&](auto kmax_c) {
        constexpr int KMAX = decltype(kmax_c)::value;
        const int pp = tid & 63, part = tid >> 6;
        const int x2 = tx * kQT + (pp & 7), y2 = ty * kQT + (pp >> 3);
        const bool in2 = (x2 < a.w) && (y2 < a.h);
        const size_t p2 = (size_t)min(y2, a.h - 1) * a.w + min(x2, a.w - 1);
        float col[KMAX];
        float m = -INFINITY;
#pragma unroll
        for (int t = 0; t < KMAX; ++t) {
            const int k = part + 4 * t;
            col[t] = (k < a.D) ? -__builtin_nontemporal_load(out + (size_t)k * hw + p2) : -INFINITY;
            m = fmaxf(m, col[t]);
        }
        red[part * 64 + pp] = m;
        __syncthreads();
        m = fmaxf(fmaxf(red[pp], red[64 + pp]), fmaxf(red[128 + pp], red[192 + pp]));
        float s = 0.f;
#pragma unroll
        for (int t = 0; t < KMAX; ++t)
            if (part + 4 * t < a.D) s += expf(col[t] - m);
        red[256 + part * 64 + pp] = s;
        __syncthreads();
        s = (red[256 + pp] + red[256 + 64 + pp]) + (red[256 + 128 + pp] + red[256 + 192 + pp]);
        const float ls = logf(s);
        if (in2) {
#pragma unroll
            for (int t = 0; t < KMAX; ++t) {
                const int k = part + 4 * t;
                if (k < a.D) a.out_logp[(size_t)k * hw + p2] = (col[t] - m) - ls;
            }
        }
    };
    if (a.D <= 64) softmax(std::integral_constant<int, 16>{}); else softmax(std::integral_constant<int, 32>{});
    CVT_END(9);
    CVT_FLUSH();
}

bool costvol_quad_supported(const CostvolArgs& a) {
    const bool extra = a.Cp == 68 && a.C > 64;
    const bool plain = a.Cp == 64 && a.C == 64;
    // 32-bit byte offsets into a view; the LDS image (patch + 20 B per candidate + accumulators) has to fit a workgroup's share
    // ... and a row pitch below 8 MB and fewer than 2^23 pixels: the address arithmetic uses __mul24, a SIGNED 24-bit multiply whose
    // operands must stay below 2^23 (a pitch in [8, 16) MB would be sign-extended to a negative offset)
    return (extra || plain) && (long)a.h * a.w * a.Cp * 4 < (1L << 31) && (long)a.w * a.Cp * 4 < (1L << 23) && (long)a.h * a.w < (1L << 23) && a.D <= NRGBD_MAX_D;
}

// Returns NRGBD_OK and sets *did_softmax when the launch also produced out_logp.
int launch_costvol_quad(const CostvolArgs& args, hipStream_t stream, bool* did_softmax) {
    CostvolArgs a = args;
    const int tiles = ceil_div(a.w, kQT) * ceil_div(a.h, kQT);
    // one workgroup per tile owns all D candidates when that fills the chip (3 workgroups per CU); smaller grids split the
    // candidates into chunks (>= 8 each) so that every CU has work, and leave the log-softmax to its own launch
    int nchunk = 1;
    while (tiles * nchunk < 3 * 256 && ceil_div(a.D, nchunk * 2) >= 8) nchunk *= 2;
    if (const int forced = dev_env_int("NRGBD_QUAD_CHUNKS")) nchunk = forced;   // developer builds only
    a.nchunk = nchunk;
    a.kchunk = ceil_div(a.D, nchunk);
    a.fuse_softmax = (nchunk == 1 && a.out_logp != nullptr && a.D <= 128) ? 1 : 0;
    *did_softmax = a.fuse_softmax != 0;
    const int patch = kQPatch3;
    const size_t lds = (size_t)patch * kQFeatBytes + (size_t)((patch + 63) / 64 * 64) * 16 + (size_t)kQPass * sizeof(int4) +
                       (size_t)((a.D + 3) & ~3) * sizeof(float) + (size_t)64 * kQAccPitch * sizeof(float);
    const dim3 grid(tiles * nchunk);
    const int tail = a.Cp == 68 ? (a.C - 64 == 3 ? 3 : -1) : 0;
    const bool al = a.align != 0;
    hipError_t e = hipSuccess;
#define NRGBD_QUAD_CFG(DIST, TL, AL)                                                                                           \
    do {                                                                                                                       \
        if (lds > 64 * 1024) {                                                                                                 \
            e = set_max_dynamic_lds(reinterpret_cast<const void*>(&costvol_quad<DIST, TL, AL, kQPatch3, 3>),                   \
                                    (int)lds);                                     \
            if (e != hipSuccess) return (int)e;                                                                                \
        }                                                                                                                      \
        hipLaunchKernelGGL((costvol_quad<DIST, TL, AL, kQPatch3, 3>), grid, dim3(256), lds, stream, a);                        \
    } while (0)
#define NRGBD_QUAD_DISPATCH(DIST)                                                                   \
    if (tail == 3) { if (al) NRGBD_QUAD_CFG(DIST, 3, true); else NRGBD_QUAD_CFG(DIST, 3, false); }        \
    else if (tail == 0) { if (al) NRGBD_QUAD_CFG(DIST, 0, true); else NRGBD_QUAD_CFG(DIST, 0, false); }   \
    else { if (al) NRGBD_QUAD_CFG(DIST, -1, true); else NRGBD_QUAD_CFG(DIST, -1, false); }
    if (a.dist == NRGBD_DIST_L2) { NRGBD_QUAD_DISPATCH(NRGBD_DIST_L2) } else { NRGBD_QUAD_DISPATCH(NRGBD_DIST_L1) }
#undef NRGBD_QUAD_DISPATCH
#undef NRGBD_QUAD_CFG
    NRGBD_CHECK_LAUNCH();
    return NRGBD_OK;
}

}  // namespace nrgbd
