// common.hpp — device helpers shared by the gfx950 kernels of libnrgbd_hip.so.
//
// Arithmetic contract (DESIGN.md §"Numerics"): the sampling coordinates are computed with the
// same operation sequence as the reference's torch code so that tap selection (floor) and
// weights agree with it to the last ulp wherever the reference itself is deterministic:
//   P = term1 + term2*d (separate mul and add; warping/homography.py:433), P /= (P_z + 1e-10)
//   (:434), g = (u - c)/c (:441-445, IEEE division), grid_sample un-normalisation
//   ((g+1)*size - 1)/2 (ATen GridSampler.h grid_sampler_unnormalize).
// The library is compiled with -ffp-contract=off; every fused multiply-add below is explicit.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/nrgbd.h"

#define NRGBD_CHECK_LAUNCH()                         \
    do {                                             \
        hipError_t e__ = hipGetLastError();          \
        if (e__ != hipSuccess) return (int)e__;      \
    } while (0)

#include <map>
#include <mutex>
#include <utility>
#ifdef NRGBD_DEV
#include <cstdlib>
#endif

namespace nrgbd {

// Developer switches (tile order experiments, prefetch ablations) exist only in -DNRGBD_DEV builds: the product library
// never reads the environment, so a stray variable cannot change what it computes.
inline int dev_env_int(const char* name) {
#ifdef NRGBD_DEV
    const char* v = getenv(name);
    return v ? atoi(v) : 0;
#else
    (void)name;
    return 0;
#endif
}

constexpr int kWave = 64;  // gfx950 wavefront

// term2 for one (pixel, view): (K R_v) * ray_p  — homography.py:317 (matmul, K=3 chain).
struct SweepTerm {
    float t1x, t1y, t1z;  // K t_v          (term1)
    float t2x, t2y, t2z;  // (K R_v) ray_p  (term2)
};

__device__ __forceinline__ SweepTerm make_sweep_term(const float* __restrict__ KR,
                                                      const float* __restrict__ Kt, float rx,
                                                      float ry, float rz) {
    SweepTerm s;
    s.t1x = Kt[0]; s.t1y = Kt[1]; s.t1z = Kt[2];
    s.t2x = __builtin_fmaf(KR[2], rz, __builtin_fmaf(KR[1], ry, KR[0] * rx));
    s.t2y = __builtin_fmaf(KR[5], rz, __builtin_fmaf(KR[4], ry, KR[3] * rx));
    s.t2z = __builtin_fmaf(KR[8], rz, __builtin_fmaf(KR[7], ry, KR[6] * rx));
    return s;
}

// ATen grid_sampler_unnormalize
__device__ __forceinline__ float unnormalize(float g, float size, bool align_corners) {
    return align_corners ? ((g + 1.f) / 2.f) * (size - 1.f) : ((g + 1.f) * size - 1.f) / 2.f;
}

// Source-image sample position (in texels) of the pixel behind `s` on the plane at depth d.
__device__ __forceinline__ void sweep_sample_pos(const SweepTerm& s, float d, float cx, float cy,
                                                 float wf, float hf, bool align_corners,
                                                 float& ix, float& iy) {
    const float px = s.t1x + s.t2x * d;
    const float py = s.t1y + s.t2y * d;
    const float pz = s.t1z + s.t2z * d;
    const float den = pz + 1e-10f;
    const float u = px / den;
    const float v = py / den;
    const float gx = (u - cx) / cx;
    const float gy = (v - cy) / cy;
    ix = unnormalize(gx, wf, align_corners);
    iy = unnormalize(gy, hf, align_corners);
}

// a / c for a loop-invariant divisor c, rc = RN(1/c) computed in double on the host: q = a rc; r = fma(-q, c, a) (exact);
// q' = fma(r, rc, q) is the correctly rounded quotient (Markstein's theorem; checked exhaustively over every finite fp32 a
// for the principal points of all configs by oracle_div_const_mismatches) — bit-identical to the IEEE division the
// reference performs, in 3 instructions instead of ~12.
__device__ __forceinline__ float div_by_const(float a, float c, float rc) {
    const float q = a * rc;
    const float r = __builtin_fmaf(-q, c, a);
    return __builtin_fmaf(r, rc, q);
}

// (nx / d, ny / d) with ONE refined reciprocal.  This is instruction for instruction the sequence the compiler emits for an
// IEEE fp32 division (v_rcp_f32, one Newton step, q0 = n r, two residual corrections) without v_div_scale / v_div_fixup,
// which are the identity unless an operand or the quotient is within a factor 2^32 of under/overflow — never the case for
// pixel coordinates (|n| < 1e7, |d| in [1e-10, 1e3]); the result is therefore the same correctly rounded quotient, in 13
// instructions for the pair instead of 24.  d = 0 / non-finite inputs give NaN or inf like the division; both select no tap.
__device__ __forceinline__ void div_pair(float nx, float ny, float d, float& qx, float& qy) {
    float r = __builtin_amdgcn_rcpf(d);
    r = __builtin_fmaf(__builtin_fmaf(-d, r, 1.0f), r, r);
    float q = nx * r;
    q = __builtin_fmaf(__builtin_fmaf(-d, q, nx), r, q);
    qx = __builtin_fmaf(__builtin_fmaf(-d, q, nx), r, q);
    q = ny * r;
    q = __builtin_fmaf(__builtin_fmaf(-d, q, ny), r, q);
    qy = __builtin_fmaf(__builtin_fmaf(-d, q, ny), r, q);
}

// sweep_sample_pos with the two divisions by the principal point done by div_by_const and the perspective division by
// div_pair (same bits as sweep_sample_pos)
template <bool ALIGN>
__device__ __forceinline__ void sweep_sample_pos_fast(const SweepTerm& s, float d, float cx, float cy, float rcx, float rcy,
                                                      float wf, float hf, float& ix, float& iy) {
    const float px = s.t1x + s.t2x * d;
    const float py = s.t1y + s.t2y * d;
    const float pz = s.t1z + s.t2z * d;
    const float den = pz + 1e-10f;
    float u, v;
    div_pair(px, py, den, u, v);
    const float gx = div_by_const(u - cx, cx, rcx);
    const float gy = div_by_const(v - cy, cy, rcy);
    ix = unnormalize(gx, wf, ALIGN);
    iy = unnormalize(gy, hf, ALIGN);
}

// sweep_sample_pos with the two divisions by the principal point done by div_by_const (same bits)
__device__ __forceinline__ void sweep_sample_pos_rc(const SweepTerm& s, float d, float cx, float cy, float rcx, float rcy,
                                                    float wf, float hf, bool align_corners, float& ix, float& iy) {
    const float px = s.t1x + s.t2x * d;
    const float py = s.t1y + s.t2y * d;
    const float pz = s.t1z + s.t2z * d;
    const float den = pz + 1e-10f;
    const float u = px / den;
    const float v = py / den;
    const float gx = div_by_const(u - cx, cx, rcx);
    const float gy = div_by_const(v - cy, cy, rcy);
    ix = unnormalize(gx, wf, align_corners);
    iy = unnormalize(gy, hf, align_corners);
}

// Bilinear footprint with zeros padding: the four corner weights (already zeroed for corners
// outside the image) and clamped integer corners, so loads are always in range.
struct Bilinear {
    int x0, x1, y0, y1;      // clamped to the image
    float nw, ne, sw, se;    // weight * in-bounds
};

__device__ __forceinline__ Bilinear bilinear_zeros(float ix, float iy, int w, int h) {
    Bilinear b;
    const float x0f = floorf(ix), y0f = floorf(iy);
    const float fx = ix - x0f, fy = iy - y0f;
    const float ex = 1.f - fx, ey = 1.f - fy;
    const float x1f = x0f + 1.f, y1f = y0f + 1.f;
    const float wm = (float)(w - 1), hm = (float)(h - 1);
    // float compares: NaN / out-of-range positions select nothing (ATen CPU mask semantics)
    const bool vx0 = (x0f >= 0.f) && (x0f <= wm), vx1 = (x1f >= 0.f) && (x1f <= wm);
    const bool vy0 = (y0f >= 0.f) && (y0f <= hm), vy1 = (y1f >= 0.f) && (y1f <= hm);
    b.x0 = vx0 ? (int)x0f : 0; b.x1 = vx1 ? (int)x1f : 0;
    b.y0 = vy0 ? (int)y0f : 0; b.y1 = vy1 ? (int)y1f : 0;
    b.nw = (vx0 && vy0) ? ey * ex : 0.f;
    b.ne = (vx1 && vy0) ? ey * fx : 0.f;
    b.sw = (vx0 && vy1) ? fy * ex : 0.f;
    b.se = (vx1 && vy1) ? fy * fx : 0.f;
    return b;
}

__device__ __forceinline__ float lerp4(float a, float b, float c, float d, const Bilinear& t) {
    float s = a * t.nw;
    s = __builtin_fmaf(b, t.ne, s);
    s = __builtin_fmaf(c, t.sw, s);
    s = __builtin_fmaf(d, t.se, s);
    return s;
}

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

inline int ceil_div(long a, long b) { return (int)((a + b - 1) / b); }

// Opt-in of a kernel for more than 64 KB of dynamic LDS (hipFuncAttributeMaxDynamicSharedMemorySize): set ONCE per (function, device),
// to the largest size the function is ever launched with — never per call.  The attribute belongs to the FUNCTION, and a launch recorded
// in a hipGraph is replayed under whatever value it has at that moment: a later eager call with a smaller size (the Winograd kernels'
// size depends on Cin) must not shrink it under a captured launch that needs more.  (Round 6 looked here first for the replay hazard
// described in neuralrgbd_amd/__init__.py; it was not the cause, the rule is kept because it is the correct use of the attribute.)
inline hipError_t set_max_dynamic_lds(const void* func, int bytes) {
    static std::mutex mu;
    static std::map<std::pair<const void*, int>, int> done;      // (function, device) -> bytes set
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    std::lock_guard<std::mutex> lk(mu);
    int& have = done[std::make_pair(func, dev)];
    if (have >= bytes) return hipSuccess;
    e = hipFuncSetAttribute(func, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e == hipSuccess) have = bytes;
    return e;
}

// Last step of every batch-statistics finaliser (conv2d.hip, conv3d.hip, wino_pc.hip): (sum, sum of squares) over `count`
// values, reduced in fp64 from the convolution epilogues' fp32 per-tile partials -> BatchNorm (scale, shift) + running statistics.
// VARIANCE-COLLAPSE GUARD.  The variance is E[y^2] - mean^2; a tile's fp32 partial of y^2 carries a relative
// error of up to ~1.5e-5 (256 sequential adds), so the computed variance is off by up to ~1.5e-5 mean^2.  When it comes out below
// kBnCollapse * mean^2 (std / |mean| < 3.2e-3) it has no correct digit left: the reference (ATen: two-pass statistics) would still
// normalise correctly, this path cannot — and the clamped-FMA ReLU of the K-Net (ops.relu_unit), whose bound assumes a computed
// variance of at least a quarter of the true one, could saturate without an error.  Such a channel gets NaN as its scale: every
// non-ReLU consumer sees NaN — and, because the kernels' ReLU (v_max_f32 / the FMA clamp) maps NaN to 0 and would hide it again,
// the finaliser also counts the channel into the caller's status word (`collapse_count`): the host mirror raises NrgbdError at its next
// synchronisation point (neuralrgbd_amd.nets.check_status: KVNET's valid_dpv probe, DepthStream.step) instead of handing out a
// wrong depth map.  No collapse reported => computed var >= 1e-5 mean^2 and true var <= computed + 1.5e-5 mean^2 <= 2.5 x computed:
// inside relu_unit's factor 4.
// (A channel that is exactly constant and non-zero would trip it too; a convolution without bias over a non-constant input has none.)
constexpr double kBnCollapse = 1e-5;
__device__ __forceinline__ void bn_finalize_channel(double sum, double sumsq, double count, float gamma, float beta, float eps,
                                                    float momentum, float* running_mean, float* running_var, float* ss, int c,
                                                    unsigned int* collapse_count) {
    const double mean = sum / count;
    double var = sumsq / count - mean * mean;
    const bool collapsed = var < kBnCollapse * mean * mean;          // false for mean == 0 (a dead channel: var == 0, scale finite)
    if (collapsed && collapse_count) atomicAdd(collapse_count, 1u);  // the caller's status word (nets.check_status raises on it)
    var = var > 0.0 ? var : 0.0;
    const float invstd = (float)(1.0 / sqrt(var + (double)eps));
    const float sc = collapsed ? __builtin_nanf("") : gamma * invstd;
    ss[2 * c] = sc;
    ss[2 * c + 1] = beta - (float)mean * sc;
    if (running_mean) {
        running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)mean;
        const double unbiased = count > 1.0 ? var * count / (count - 1.0) : var;
        running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unbiased;
    }
}

// ---------------------------------------------------------------------------------------------------------------------
#define NRGBD_EXP_INF __builtin_inff()
#define NRGBD_EXP_RINT __builtin_rint
__device__ __forceinline__ float exp_rn(float xf) {
    /* exp(x) for the byte-exact export: evaluated in fp64 by a written-out sequence (round-to-nearest-even range reduction
     * by ln 2 in two parts, degree-13 Taylor polynomial in Horner form with separately rounded products and sums, exact
     * scaling by 2^k) and rounded once to fp32 — the correctly rounded expf up to double rounding (~1 input in 2^29).  The
     * device kernel and the CPU oracle execute the SAME operations, so the uint16 maps agree bit for bit; libm / the device's
     * expf are 1-ulp functions that differ from each other in ~5 % of inputs, which after `(map * 1000).astype(uint16)`
     * flips the last unit of ~1e-4 of the pixels. */
    const double x = (double)xf;
    if (!(x > -104.0)) return (x != x) ? xf : 0.0f;          /* below the smallest fp32 subnormal / NaN */
    if (x > 88.8) return NRGBD_EXP_INF;
    const double kd = NRGBD_EXP_RINT(x * 1.4426950408889634074);
    double r = x - kd * 0.693147180369123816490;             /* ln2 high part: 32 significant bits, kd*hi is exact */
    r = r - kd * 1.90821492927058770002e-10;                 /* ln2 low part */
    double p = 1.6059043836821613e-10;                       /* 1/13! */
    p = p * r + 2.08767569878681e-09;                        /* 1/12! */
    p = p * r + 2.505210838544172e-08;                       /* 1/11! */
    p = p * r + 2.755731922398589e-07;                       /* 1/10! */
    p = p * r + 2.7557319223985893e-06;                      /* 1/9!  */
    p = p * r + 2.48015873015873e-05;                        /* 1/8!  */
    p = p * r + 0.0001984126984126984;                       /* 1/7!  */
    p = p * r + 0.001388888888888889;                        /* 1/6!  */
    p = p * r + 0.008333333333333333;                        /* 1/5!  */
    p = p * r + 0.041666666666666664;                        /* 1/4!  */
    p = p * r + 0.16666666666666666;                         /* 1/3!  */
    p = p * r + 0.5;
    p = p * r + 1.0;
    p = p * r + 1.0;
    const long long k = (long long)kd;
    union { unsigned long long u; double d; } s;
    s.u = (unsigned long long)(k + 1023) << 52;              /* 2^k, k in [-151, 129]: a normal double */
    return (float)(p * s.d);
}
#undef NRGBD_EXP_INF
#undef NRGBD_EXP_RINT

}  // namespace nrgbd
