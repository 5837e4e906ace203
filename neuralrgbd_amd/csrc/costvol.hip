// costvol.hip — fused plane-sweep cost volume for gfx950:
//   homography (K1) + bilinear gather (K2) + channel reduction / view accumulate (K3)
//   + log-softmax over depth (K4) in ONE launch.
// Replaces warping/homography.py:293-331,421-448,81-87 and models/basic.py:299-300, which in
// the reference materialise a [D,C,h,w] warped tensor per view (843 MB at the 192x256x64 grid)
// and re-read it ~8x; here nothing but the cost / log-prob volume is written.
//
// Kernel `costvol_gather` (generation 1): one lane = one reference pixel of a TW x TH tile and
// one of KS depth sub-groups; the 4 waves of a workgroup split the depth candidates of the same
// tile, so the reference texel (held in registers) and the tile's source footprint are shared
// through L1/L2.  Source texels are gathered straight from the NHWC tensor with 16-byte loads.
// The per-candidate costs of the tile are parked in LDS so that the log-softmax over all D
// candidates is taken in the same launch.
#include <cstdlib>

#include "costvol.hpp"

namespace nrgbd {


// CP4 = Cp/4 when the reference texel is cached in registers, 0 = generic (re-read per use).
template <int TW, int TH, int KS, int CP4>
__global__ __launch_bounds__(256) void costvol_gather(const CostvolArgs a) {
    static_assert(TW * TH * KS == 64, "one wave = TW x TH pixels x KS depth sub-groups");
    constexpr int TPX = TW * TH;
    constexpr int NKG = 4 * KS;  // depth groups per workgroup
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* cost_s = lds;                          // [D][TPX]
    float* red_s = lds + (size_t)a.D * TPX;       // [NKG][TPX] scratch for the softmax

    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int pix = lane % TPX, ks = lane / TPX;
    const int tiles_x = (a.w + TW - 1) / TW;
    const int tx = blockIdx.x % tiles_x, ty = blockIdx.x / tiles_x;
    const int x = tx * TW + pix % TW, y = ty * TH + pix / TW;
    const bool inside = (x < a.w) && (y < a.h);
    const int xc = min(x, a.w - 1), yc = min(y, a.h - 1);
    const size_t hw = (size_t)a.h * a.w;
    const size_t p = (size_t)yc * a.w + xc;
    const int cp4 = a.Cp >> 2;

    const float rx = a.rays[p], ry = a.rays[hw + p], rz = a.rays[2 * hw + p];
    const float4* refp = reinterpret_cast<const float4*>(a.ref + p * a.Cp);
    float4 rreg[CP4 > 0 ? CP4 : 1];
    if (CP4 > 0) {
#pragma unroll
        for (int i = 0; i < CP4; ++i) rreg[i] = refp[i];
    }
    // channels >= C of the last 16-B word do not take part in the distance
    const int tail = a.C - 4 * (cp4 - 1);  // 1..4 valid components in the last word
    const float wf = (float)a.w, hf = (float)a.h;

    const int kg = wv * KS + ks;
    const int kpg = (a.D + NKG - 1) / NKG;
    const int k_begin = kg * kpg, k_end = min(a.D, k_begin + kpg);

    for (int k = k_begin; k < k_end; ++k) {
        const float d = a.d_candi[k];
        float total = 0.f;
        for (int v = 0; v < a.V; ++v) {
            const SweepTerm st = make_sweep_term(a.KR + 9 * v, a.Kt + 3 * v, rx, ry, rz);
            float ix, iy;
            sweep_sample_pos(st, d, a.cx, a.cy, wf, hf, a.align != 0, ix, iy);
            const Bilinear b = bilinear_zeros(ix, iy, a.w, a.h);
            const float* sv = a.src + (size_t)v * hw * a.Cp;
            const float4* pnw = reinterpret_cast<const float4*>(sv + ((size_t)b.y0 * a.w + b.x0) * a.Cp);
            const float4* pne = reinterpret_cast<const float4*>(sv + ((size_t)b.y0 * a.w + b.x1) * a.Cp);
            const float4* psw = reinterpret_cast<const float4*>(sv + ((size_t)b.y1 * a.w + b.x0) * a.Cp);
            const float4* pse = reinterpret_cast<const float4*>(sv + ((size_t)b.y1 * a.w + b.x1) * a.Cp);
            float acc = 0.f;
            auto word = [&](int i, int ncomp) {
                const float4 A = pnw[i], B = pne[i], Cc = psw[i], Dd = pse[i];
                float4 r;
                if constexpr (CP4 > 0) r = rreg[i]; else r = refp[i];
                const float s0 = lerp4(A.x, B.x, Cc.x, Dd.x, b) - r.x;
                const float s1 = lerp4(A.y, B.y, Cc.y, Dd.y, b) - r.y;
                const float s2 = lerp4(A.z, B.z, Cc.z, Dd.z, b) - r.z;
                const float s3 = lerp4(A.w, B.w, Cc.w, Dd.w, b) - r.w;
                if (a.dist == NRGBD_DIST_L2) {
                    acc = __builtin_fmaf(s0, s0, acc);
                    if (ncomp > 1) acc = __builtin_fmaf(s1, s1, acc);
                    if (ncomp > 2) acc = __builtin_fmaf(s2, s2, acc);
                    if (ncomp > 3) acc = __builtin_fmaf(s3, s3, acc);
                } else {
                    acc += fabsf(s0);
                    if (ncomp > 1) acc += fabsf(s1);
                    if (ncomp > 2) acc += fabsf(s2);
                    if (ncomp > 3) acc += fabsf(s3);
                }
            };
            if (CP4 > 0) {
#pragma unroll
                for (int i = 0; i < CP4 - 1; ++i) word(i, 4);
                word(CP4 - 1, tail);
            } else {
                for (int i = 0; i < cp4 - 1; ++i) word(i, 4);
                word(cp4 - 1, tail);
            }
            total = total + acc / a.sigma;  // homography.py:325, views in order
        }
        cost_s[k * TPX + pix] = total;
        if (a.out_cost && inside) a.out_cost[(size_t)k * hw + p] = total;
    }
    if (!a.out_logp) return;
    __syncthreads();

    // log_softmax(-cost) over all D candidates of each tile pixel (models/basic.py:299-300):
    // thread (pix, g) covers k = g, g+NKG, ...
    const int g = threadIdx.x / TPX;  // 0..NKG-1 (256 threads = NKG * TPX)
    const int pp = threadIdx.x % TPX;
    float m = -INFINITY;
    for (int k = g; k < a.D; k += NKG) m = fmaxf(m, -cost_s[k * TPX + pp]);
    red_s[g * TPX + pp] = m;
    __syncthreads();
    m = red_s[pp];
    for (int j = 1; j < NKG; ++j) m = fmaxf(m, red_s[j * TPX + pp]);
    __syncthreads();
    float s = 0.f;
    for (int k = g; k < a.D; k += NKG) s += expf(-cost_s[k * TPX + pp] - m);
    red_s[g * TPX + pp] = s;
    __syncthreads();
    s = red_s[pp];
    for (int j = 1; j < NKG; ++j) s += red_s[j * TPX + pp];
    const float ls = logf(s);
    const int x2 = tx * TW + pp % TW, y2 = ty * TH + pp / TW;
    if (x2 < a.w && y2 < a.h) {
        const size_t p2 = (size_t)y2 * a.w + x2;
        for (int k = g; k < a.D; k += NKG) a.out_logp[(size_t)k * hw + p2] = (-cost_s[k * TPX + pp] - m) - ls;
    }
}

template <int TW, int TH, int KS>
static int launch_gather(const CostvolArgs& a, hipStream_t stream) {
    constexpr int TPX = TW * TH, NKG = 4 * KS;
    const int tiles = ceil_div(a.w, TW) * ceil_div(a.h, TH);
    const size_t lds = ((size_t)a.D + NKG) * TPX * sizeof(float);
    const int cp4 = a.Cp >> 2;
#define NRGBD_GATHER_CASE(N)                                                                     \
    case N:                                                                                      \
        hipLaunchKernelGGL((costvol_gather<TW, TH, KS, N>), dim3(tiles), dim3(256), lds, stream, a); \
        break;
    switch (cp4) {
        NRGBD_GATHER_CASE(1)
        NRGBD_GATHER_CASE(2)
        NRGBD_GATHER_CASE(3)
        NRGBD_GATHER_CASE(4)
        NRGBD_GATHER_CASE(8)
        NRGBD_GATHER_CASE(9)
        NRGBD_GATHER_CASE(16)
        NRGBD_GATHER_CASE(17)
        default:
            hipLaunchKernelGGL((costvol_gather<TW, TH, KS, 0>), dim3(tiles), dim3(256), lds, stream, a);
    }
#undef NRGBD_GATHER_CASE
    NRGBD_CHECK_LAUNCH();
    return NRGBD_OK;
}

}  // namespace nrgbd

extern "C" int nrgbd_costvol_fwd_gen(const float* ref_nhwc, const float* src_nhwc, const float* KR,
                                     const float* Kt, const float* rays, const float* d_candi,
                                     float cx, float cy, float sigma, int dist, int align_corners,
                                     float* out_cost, float* out_logp, int V, int C, int Cp, int D,
                                     int h, int w, int generation, void* stream) {
    using namespace nrgbd;
    if (!ref_nhwc || !src_nhwc || !KR || !Kt || !rays || !d_candi) return NRGBD_E_NULL;
    if (!out_cost && !out_logp) return NRGBD_E_NULL;
    if (V <= 0 || V > NRGBD_MAX_V || C <= 0 || D <= 0 || D > NRGBD_MAX_D || h <= 0 || w <= 0) return NRGBD_E_SHAPE;
    if ((Cp & 3) || Cp < C || Cp - C > 3) return NRGBD_E_ALIGN;
    if ((reinterpret_cast<uintptr_t>(ref_nhwc) | reinterpret_cast<uintptr_t>(src_nhwc)) & 15) return NRGBD_E_ALIGN;
    if (dist != NRGBD_DIST_L2 && dist != NRGBD_DIST_L1) return NRGBD_E_ARG;
    if (generation < NRGBD_GEN_AUTO || generation > NRGBD_GEN_QUAD) return NRGBD_E_ARG;
    CostvolArgs a{ref_nhwc, src_nhwc, KR, Kt, rays, d_candi, out_cost, out_logp, cx, cy, sigma,
                  dist, align_corners, V, C, Cp, D, h, w, 0, 0, 1, D, 0,
                  (float)(1.0 / (double)cx), (float)(1.0 / (double)cy), (float)(1.0 / (double)sigma)};
    a.trace = nullptr;
#ifdef NRGBD_DEV
    a.debug = dev_env_int("NRGBD_ABLATE");
    if (const char* tp = getenv("NRGBD_CV_TRACE")) a.trace = reinterpret_cast<long long*>(strtoull(tp, nullptr, 0));
#endif
    hipStream_t s = (hipStream_t)stream;
    // AUTO: generation 3 (quad) for the path's own texel (64 feature channels [+ RGB word]); generation 2 (LDS, lane =
    // pixel) for the other channel counts it instantiates; generation 1 (gather) for everything else.  An explicit
    // generation that does not support the shape is an error, never a silent substitution.
    if (generation == NRGBD_GEN_QUAD && !costvol_quad_supported(a)) return NRGBD_E_SHAPE;
    if (generation == NRGBD_GEN_LDS && !costvol_lds_supported(Cp >> 2)) return NRGBD_E_SHAPE;
    if (generation == NRGBD_GEN_QUAD || (generation == NRGBD_GEN_AUTO && costvol_quad_supported(a))) {
        bool did_softmax = false;
        int rc = launch_costvol_quad(a, s, &did_softmax);
        if (rc != NRGBD_OK) return rc;
        if (out_logp && !did_softmax)
            return launch_logsoftmax_d(out_cost ? out_cost : out_logp, nullptr, -1.f, out_logp, D, (size_t)h * w, s);
        return NRGBD_OK;
    }
    if (generation == NRGBD_GEN_LDS || (generation == NRGBD_GEN_AUTO && costvol_lds_supported(Cp >> 2))) {
        int rc = launch_costvol_lds(a, s);
        if (rc != NRGBD_OK) return rc;
        if (out_logp)  // log_softmax(-cost) over D (models/basic.py:299-300); in place when only logp is wanted
            return launch_logsoftmax_d(out_cost ? out_cost : out_logp, nullptr, -1.f, out_logp, D, (size_t)h * w, s);
        return NRGBD_OK;
    }
    // Enough workgroups to cover 256 CUs: small grids use 16-pixel tiles with 4 depth sub-groups
    // per wave, large grids 64-pixel tiles.
    const long tiles64 = (long)ceil_div(w, 16) * ceil_div(h, 4);
    if (tiles64 >= 512) return launch_gather<16, 4, 1>(a, s);
    return launch_gather<8, 2, 4>(a, s);
}

extern "C" int nrgbd_costvol_fwd(const float* ref_nhwc, const float* src_nhwc, const float* KR,
                                 const float* Kt, const float* rays, const float* d_candi,
                                 float cx, float cy, float sigma, int dist, int align_corners,
                                 float* out_cost, float* out_logp, int V, int C, int Cp, int D,
                                 int h, int w, void* stream) {
    return nrgbd_costvol_fwd_gen(ref_nhwc, src_nhwc, KR, Kt, rays, d_candi, cx, cy, sigma, dist, align_corners, out_cost,
                                 out_logp, V, C, Cp, D, h, w, nrgbd::NRGBD_GEN_AUTO, stream);
}
