/*
 * nrgbd_oracle.c — CPU restatement of the sampling arithmetic of NVlabs/neuralrgbd's
 * plane-sweep depth path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may call it; neuralrgbd_amd/ never imports it.
 *
 * Parity pin: the reference ships no golden vectors or tests for this path (SURVEY.md §4,
 * §8c).  The pin is the reference code itself run under torch 2.10 (CPU) in the build
 * container: oracle/gen_golden.py imports the unmodified reference and writes
 * tests/golden/ (.npz); tests/test_oracle.py checks every function below against them.
 *
 * Plain scalar fp32, one rounding per written operation (build with -ffp-contract=off),
 * every function cites the reference lines it restates (paths relative to
 * /root/reference/code).  The bilinear / trilinear sampling follows torch's
 * F.grid_sample(mode='bilinear') as installed (ATen GridSampler.h:
 * grid_sampler_unnormalize / clip_coordinates; CPU kernels GridSamplerKernel.cpp and
 * GridSampler.cpp).
 */
#include <math.h>
#include <stddef.h>
#include <stdlib.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* thread control for the cpu_baseline leg of bench.py */
int oracle_set_threads(int n) {
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
    return omp_get_max_threads();
#else
    (void)n; return 1;
#endif
}

/* warping/homography.py:315-317: term1 = IntM.matmul(t_v), left factor IntM.matmul(R_v) of term2.  The order of these
 * K = 3 contractions is the one torch 2.10's CPU kernels execute in the build container where the golden vectors were
 * generated (checked against the live reference by tests/test_oracle_vs_reference.py): sgemm = fma chain over k,
 * 3-element sgemv = (p1 + p2) + p0 with separately rounded products.  Written out so that it is the same on every host. */
int oracle_homography_terms(const float* K, const float* R, const float* t, int V, float* KR, float* Kt) {
    for (int v = 0; v < V; ++v) {
        for (int r = 0; r < 3; ++r) {
            for (int c = 0; c < 3; ++c) {
                float s = K[3 * r] * R[9 * v + c];
                s = fmaf(K[3 * r + 1], R[9 * v + 3 + c], s);
                s = fmaf(K[3 * r + 2], R[9 * v + 6 + c], s);
                KR[9 * v + 3 * r + c] = s;
            }
            float p0 = K[3 * r] * t[3 * v], p1 = K[3 * r + 1] * t[3 * v + 1], p2 = K[3 * r + 2] * t[3 * v + 2];
            Kt[3 * v + r] = (p1 + p2) + p0;
        }
    }
    return 0;
}

/* test_utils/test_KVNet.py:50,52: `Src_CamPoses[ibatch, t_win_r].inverse()` — the motion the PREDICT step resamples through.
 * The reference leaves the operation order to the host LAPACK (MKL sgetrf + sgetrs on the transposed matrix under torch
 * 2.10; its LU is reproducible as a right-looking fma chain with reciprocal scaling, its triangular solves are not), so
 * the path fixes it: Gauss-Jordan with partial pivoting on [A | I] in fp64, one rounding per operation, rounded to fp32.
 * Same sequence as neuralrgbd_amd/csrc/geom.hip::pose_inverse_kernel.  tests/test_oracle_vs_reference.py bounds the
 * distance to the live reference's `.inverse()` (a few fp32 ulps = the reference's own rounding error). */
int oracle_pose_inverse(const float* T, float* out, int n) {
    int singular = 0;
    for (int m = 0; m < n; ++m) {
        double a[4][8];
        for (int i = 0; i < 4; ++i)
            for (int j = 0; j < 4; ++j) { a[i][j] = (double)T[16 * m + 4 * i + j]; a[i][4 + j] = (i == j) ? 1.0 : 0.0; }
        int bad = 0;
        for (int c = 0; c < 4 && !bad; ++c) {
            int p = c;
            double best = fabs(a[c][c]);
            for (int r = c + 1; r < 4; ++r) { double v = fabs(a[r][c]); if (v > best) { best = v; p = r; } }
            if (!(best > 0.0)) { bad = 1; break; }
            if (p != c) for (int j = 0; j < 8; ++j) { double tmp = a[c][j]; a[c][j] = a[p][j]; a[p][j] = tmp; }
            double piv = a[c][c];
            for (int j = 0; j < 8; ++j) a[c][j] = a[c][j] / piv;
            for (int r = 0; r < 4; ++r) {
                if (r == c) continue;
                double f = a[r][c];
                for (int j = 0; j < 8; ++j) a[r][j] = a[r][j] - f * a[c][j];
            }
        }
        singular += bad;
        for (int i = 0; i < 4; ++i)
            for (int j = 0; j < 4; ++j) out[16 * m + 4 * i + j] = bad ? NAN : (float)a[i][4 + j];
    }
    return singular;
}

/* ATen GridSampler.h grid_sampler_unnormalize */
static inline float unnormalize(float g, int size, int align_corners) {
    if (align_corners) return ((g + 1.f) / 2.f) * (float)(size - 1);
    return ((g + 1.f) * (float)size - 1.f) / 2.f;
}

/* warping/homography.py:315-317 (term1 = K t, term2 = (K R) ray) and :433-445
 * (P = term1 + term2*d; P /= (P_z + 1e-10); (u - cx)/cx, (v - cy)/cy). */
static inline void sweep_coords(const float* KR, const float* Kt, float rx, float ry,
                                float rz, float d, float cx, float cy, int w, int h,
                                int align_corners, float* ix, float* iy) {
    float t2x = KR[0] * rx; t2x = fmaf(KR[1], ry, t2x); t2x = fmaf(KR[2], rz, t2x);
    float t2y = KR[3] * rx; t2y = fmaf(KR[4], ry, t2y); t2y = fmaf(KR[5], rz, t2y);
    float t2z = KR[6] * rx; t2z = fmaf(KR[7], ry, t2z); t2z = fmaf(KR[8], rz, t2z);
    float px = Kt[0] + t2x * d;
    float py = Kt[1] + t2y * d;
    float pz = Kt[2] + t2z * d;
    float den = pz + 1e-10f;
    float u = px / den, v = py / den;
    float gx = (u - cx) / cx, gy = (v - cy) / cy;
    *ix = unnormalize(gx, w, align_corners);
    *iy = unnormalize(gy, h, align_corners);
}

typedef struct { int x[2], y[2]; float wx[2], wy[2]; int vx[2], vy[2]; } taps2d;

/* F.grid_sample bilinear, padding_mode='zeros' (homography.py:447): floor, weights
 * (1-w, w), a tap contributes only when inside [0,W-1]x[0,H-1]. */
static inline void bilinear_taps(float ix, float iy, int w, int h, taps2d* t) {
    float x0 = floorf(ix), y0 = floorf(iy);
    float fx = ix - x0, fy = iy - y0;
    t->wx[0] = 1.f - fx; t->wx[1] = fx;
    t->wy[0] = 1.f - fy; t->wy[1] = fy;
    for (int i = 0; i < 2; ++i) {
        float xf = x0 + (float)i, yf = y0 + (float)i;
        t->vx[i] = (xf >= 0.f && xf <= (float)(w - 1));
        t->vy[i] = (yf >= 0.f && yf <= (float)(h - 1));
        t->x[i] = t->vx[i] ? (int)xf : 0;
        t->y[i] = t->vy[i] ? (int)yf : 0;
    }
}

/*
 * Plane-sweep cost volume.  warping/homography.py:293-331 (est_swp_volume_v4),
 * :421-448 (_back_warp_homo_parallel), :81-87 (img_dis_L2_pard / L1).
 * feat_ref [C][h][w], feat_src [V][C][h][w]  (NCHW, as the reference holds them),
 * KR [V][9], Kt [V][3], rays [3][hw], cost [D][h][w].
 */
int oracle_costvol(const float* feat_ref, const float* feat_src, const float* KR,
                   const float* Kt, const float* rays, const float* d_candi, float cx,
                   float cy, float sigma, int dist, int align_corners, int V, int C,
                   int D, int h, int w, float* cost) {
    const size_t hw = (size_t)h * w;
    /* The reference loops views outermost (:313) and adds each view's [D,h,w] slab to costV
     * (:325); per output element that is cost = ((0 + S_0/sigma) + S_1/sigma) + ..., which is
     * what the v-innermost loop below computes (element-wise identical, thread-parallel). */
#pragma omp parallel for collapse(2) schedule(static)
    for (int k = 0; k < D; ++k) {
        for (size_t p = 0; p < hw; ++p) {
            float total = 0.f;                                           /* :306 */
            for (int v = 0; v < V; ++v) {                                /* :313 */
                const float* src = feat_src + (size_t)v * C * hw;
                float ix, iy; taps2d t;
                sweep_coords(KR + 9 * v, Kt + 3 * v, rays[p], rays[hw + p],
                             rays[2 * hw + p], d_candi[k], cx, cy, w, h, align_corners,
                             &ix, &iy);
                bilinear_taps(ix, iy, w, h, &t);
                float nw = t.wy[0] * t.wx[0], ne = t.wy[0] * t.wx[1];
                float sw = t.wy[1] * t.wx[0], se = t.wy[1] * t.wx[1];
                int vnw = t.vy[0] && t.vx[0], vne = t.vy[0] && t.vx[1];
                int vsw = t.vy[1] && t.vx[0], vse = t.vy[1] && t.vx[1];
                size_t onw = (size_t)t.y[0] * w + t.x[0], one = (size_t)t.y[0] * w + t.x[1];
                size_t osw = (size_t)t.y[1] * w + t.x[0], ose = (size_t)t.y[1] * w + t.x[1];
                float acc = 0.f;
                for (int c = 0; c < C; ++c) {
                    const float* pl = src + (size_t)c * hw;
                    float s = (vnw ? pl[onw] : 0.f) * nw + (vne ? pl[one] : 0.f) * ne
                            + (vsw ? pl[osw] : 0.f) * sw + (vse ? pl[ose] : 0.f) * se;
                    float df = s - feat_ref[(size_t)c * hw + p];
                    acc += (dist == 0) ? df * df : fabsf(df);            /* :81-87 */
                }
                total = total + acc / sigma;                             /* :325 */
            }
            cost[(size_t)k * hw + p] = total;
        }
    }
    return 0;
}

/*
 * K-Net input warp with the samples kept.  warping/homography.py:234-280
 * (warp_img_feats_v3; result transposed to [C][D][h][w], :261).
 * src [V][Cs][h][w] -> out [V][Cs][D][h][w].
 */
int oracle_warp_volume(const float* src, const float* KR, const float* Kt,
                       const float* rays, const float* d_candi, float cx, float cy,
                       int align_corners, int V, int Cs, int D, int h, int w, float* out) {
    const size_t hw = (size_t)h * w;
#pragma omp parallel for collapse(2) schedule(static)
    for (int v = 0; v < V; ++v)
        for (int k = 0; k < D; ++k)
            for (size_t p = 0; p < hw; ++p) {
                float ix, iy; taps2d t;
                sweep_coords(KR + 9 * v, Kt + 3 * v, rays[p], rays[hw + p],
                             rays[2 * hw + p], d_candi[k], cx, cy, w, h, align_corners,
                             &ix, &iy);
                bilinear_taps(ix, iy, w, h, &t);
                float nw = t.wy[0] * t.wx[0], ne = t.wy[0] * t.wx[1];
                float sw = t.wy[1] * t.wx[0], se = t.wy[1] * t.wx[1];
                for (int c = 0; c < Cs; ++c) {
                    const float* pl = src + ((size_t)v * Cs + c) * hw;
                    float a = (t.vy[0] && t.vx[0]) ? pl[(size_t)t.y[0] * w + t.x[0]] : 0.f;
                    float b = (t.vy[0] && t.vx[1]) ? pl[(size_t)t.y[0] * w + t.x[1]] : 0.f;
                    float cc = (t.vy[1] && t.vx[0]) ? pl[(size_t)t.y[1] * w + t.x[0]] : 0.f;
                    float dd = (t.vy[1] && t.vx[1]) ? pl[(size_t)t.y[1] * w + t.x[1]] : 0.f;
                    out[(((size_t)v * Cs + c) * D + k) * hw + p] = a * nw + b * ne + cc * sw + dd * se;
                }
            }
    return 0;
}

/* ATen GridSampler.h clip_coordinates (padding_mode='border') */
static inline float clipf(float x, int size) {
    float hi = (float)(size - 1);
    x = (x < 0.f) ? 0.f : x;        /* std::max(in, 0)          */
    return (x < hi) ? x : hi;       /* std::min(size-1, .): NaN -> size-1 */
}

/* warping/homography.py:873-887 _set_vol_border: the 6 faces read as border_val */
static inline float vol_at(const float* vol, int D, int h, int w, int z, int y, int x,
                           float pad) {
    if (z == 0 || y == 0 || x == 0 || z == D - 1 || y == h - 1 || x == w - 1) return pad;
    return vol[((size_t)z * h + y) * w + x];
}

/*
 * PREDICT: rigid 3-D resample of the DPV.  warping/homography.py:654-723
 * (resample_vol_cuda with d_candi_new=None), :873-887 (_set_vol_border) and the clamp of
 * test_utils/test_KVNet.py:54-59.  T is the row-major 4x4 rel_extM.
 */
int oracle_dpv_resample_to(const float* dpv, const float* T, const float* rays,
                           const float* d_candi, float tan_hh, float tan_hv, float z_half,
                           float z_radius, float pad, int do_clamp, float lo, float hi,
                           int D, int D_out, int h, int w, float* out) {
    /* d_candi: depths of the D_out OUTPUT planes (= the source's candidates, or d_candi_new: homography.py:675-682);
     * D: planes of the source volume dpv */
    const size_t hw = (size_t)h * w;
#pragma omp parallel for schedule(static)
    for (int k = 0; k < D_out; ++k)
        for (size_t p = 0; p < hw; ++p) {
            /* :679-682  X = d * ray */
            float d = d_candi[k];
            float X = d * rays[p], Y = d * rays[hw + p], Z = d * rays[2 * hw + p];
            /* :698-702  [4x4] matmul [X Y Z 1] */
            float q[4];
            for (int r = 0; r < 4; ++r) {
                float a = T[4 * r] * X;
                a = fmaf(T[4 * r + 1], Y, a);
                a = fmaf(T[4 * r + 2], Z, a);
                a = fmaf(T[4 * r + 3], 1.f, a);
                q[r] = a;
            }
            /* :705-707 */
            float gx = q[0] / (q[2] + 1e-10f) / tan_hh;
            float gy = q[1] / (q[2] + 1e-10f) / tan_hv;
            float gz = (q[2] - z_half) / z_radius;
            /* :710 */
            float wq = q[3] + 1e-10f;
            gx = gx / wq; gy = gy / wq; gz = gz / wq;
            /* :716 grid_sample 3-D, bilinear, border, align_corners=False */
            float fx = clipf(unnormalize(gx, w, 0), w);
            float fy = clipf(unnormalize(gy, h, 0), h);
            float fz = clipf(unnormalize(gz, D, 0), D);
            float x0f = floorf(fx), y0f = floorf(fy), z0f = floorf(fz);
            int x0 = (int)x0f, y0 = (int)y0f, z0 = (int)z0f;
            int x1 = x0 + 1, y1 = y0 + 1, z1 = z0 + 1;
            float ex = (x0f + 1.f) - fx, ey = (y0f + 1.f) - fy, ez = (z0f + 1.f) - fz;
            float wx_ = fx - x0f, wy_ = fy - y0f, wz_ = fz - z0f;
            /* GridSampler.cpp grid_sampler_3d_cpu_impl: tnw tne tsw tse bnw bne bsw bse */
            float wt[8] = { ex * ey * ez, wx_ * ey * ez, ex * wy_ * ez, wx_ * wy_ * ez,
                            ex * ey * wz_, wx_ * ey * wz_, ex * wy_ * wz_, wx_ * wy_ * wz_ };
            int xs[8] = { x0, x1, x0, x1, x0, x1, x0, x1 };
            int ys[8] = { y0, y0, y1, y1, y0, y0, y1, y1 };
            int zs[8] = { z0, z0, z0, z0, z1, z1, z1, z1 };
            float acc = 0.f;
            for (int i = 0; i < 8; ++i)
                if (xs[i] >= 0 && xs[i] < w && ys[i] >= 0 && ys[i] < h && zs[i] >= 0 && zs[i] < D)
                    acc += vol_at(dpv, D, h, w, zs[i], ys[i], xs[i], pad) * wt[i];
            if (do_clamp) { acc = acc < lo ? lo : acc; acc = acc > hi ? hi : acc; }
            out[(size_t)k * hw + p] = acc;
        }
    return 0;
}

int oracle_dpv_resample(const float* dpv, const float* T, const float* rays,
                        const float* d_candi, float tan_hh, float tan_hv, float z_half,
                        float z_radius, float pad, int do_clamp, float lo, float hi,
                        int D, int h, int w, float* out) {
    return oracle_dpv_resample_to(dpv, T, rays, d_candi, tan_hh, tan_hv, z_half, z_radius, pad, do_clamp, lo, hi, D, D, h, w, out);
}

/* models/basic.py:299-300 and models/KVNET.py:172-173: log_softmax over D of scale*a + b */
int oracle_logsoftmax_d(const float* a, const float* b, float scale, int D, size_t n,
                        float* out) {
#pragma omp parallel for schedule(static)
    for (size_t p = 0; p < n; ++p) {
        float m = -INFINITY;
        for (int k = 0; k < D; ++k) {
            float v = scale * a[(size_t)k * n + p];
            if (b) v = v + b[(size_t)k * n + p];
            m = v > m ? v : m;
        }
        float s = 0.f;
        for (int k = 0; k < D; ++k) {
            float v = scale * a[(size_t)k * n + p];
            if (b) v = v + b[(size_t)k * n + p];
            s += expf(v - m);
        }
        float ls = logf(s);
        for (int k = 0; k < D; ++k) {
            float v = scale * a[(size_t)k * n + p];
            if (b) v = v + b[(size_t)k * n + p];
            out[(size_t)k * n + p] = (v - m) - ls;
        }
    }
    return 0;
}

/* mutils/misc.py:532-548 depth_val_regression (BV_log=True); conf = max_d (export_res.py:58-59) */
int oracle_depth_regress(const float* logp, const float* d_candi, int D, size_t n,
                         float* depth, float* conf) {
    for (size_t p = 0; p < n; ++p) {
        float acc = 0.f, m = -INFINITY;
        for (int k = 0; k < D; ++k) {
            float v = logp[(size_t)k * n + p];
            acc = acc + expf(v) * d_candi[k];
            m = v > m ? v : m;
        }
        if (depth) depth[p] = acc;
        if (conf) conf[p] = m;
    }
    return 0;
}

/* models/basic.py:254-263 / models/KVNET.py:149-151: F.avg_pool2d(img, pool) */
int oracle_avgpool(const float* img, int N, int C, int H, int W, int pool, float* out) {
    int h = H / pool, w = W / pool;
    for (int nc = 0; nc < N * C; ++nc)
        for (int y = 0; y < h; ++y)
            for (int x = 0; x < w; ++x) {
                float s = 0.f;
                for (int j = 0; j < pool; ++j)
                    for (int i = 0; i < pool; ++i)
                        s += img[((size_t)nc * H + (y * pool + j)) * W + (x * pool + i)];
                out[((size_t)nc * h + y) * w + x] = s / (float)(pool * pool);
            }
    return 0;
}

/* test_utils/export_res.py:43-75 export_res_img: expected depth (sum_k exp(BV_k) d_k), confidence exp(max_k BV_k) and the
 * two uint16 maps the .pgm files hold: (map * scale).astype(np.uint16) = truncation toward zero of the fp32 product
 * (values outside [0, 65535], which the reference's ranges never produce, are clamped here to keep the cast defined). */
static inline unsigned short to_u16(float v) {
    if (!(v > 0.f)) return 0;
    if (v >= 65535.f) return 65535;
    return (unsigned short)v;
}
#define NRGBD_EXP_INF INFINITY
#define NRGBD_EXP_RINT rint
static inline float exp_rn(float xf) {
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

int oracle_export_depth_u16(const float* logp, const float* d_candi, int D, size_t n, float depth_scale,
                            float conf_scale, float* depth, float* conf, unsigned short* depth_u16,
                            unsigned short* conf_u16) {
    for (size_t p = 0; p < n; ++p) {
        float acc = 0.f, m = -INFINITY;
        for (int k = 0; k < D; ++k) {
            float v = logp[(size_t)k * n + p];
            acc = acc + exp_rn(v) * d_candi[k];
            m = v > m ? v : m;
        }
        float c = exp_rn(m);
        if (depth) depth[p] = acc;
        if (conf) conf[p] = c;
        if (depth_u16) depth_u16[p] = to_u16(acc * depth_scale);
        if (conf_u16) conf_u16[p] = to_u16(c * conf_scale);
    }
    return 0;
}

/*
 * warping/homography.py:479-528 back_warp_th_Rt_msrc (and :530-575 back_warp_th_Rt, its single-view form): warp N source
 * images to the reference view through a per-pixel DEPTH MAP (not planes) and the rigid motions (R_n, t_n); caller
 * ICP/opt_pose_numerical.py:245-294 (local bundle adjustment) differentiates the result w.r.t. R_n and t_n.
 *   X = dmap[p] * ray_p;  Y = R_n X + t_n (the 4x4 [R t; 0 1] matmul: fma chain over k = 0..3);  P = K Y (4x4 with a zero
 *   last row/column: fma chain over k = 0..2, the k = 3 term adds 0);  P /= P_z (no epsilon here, :510);
 *   g = (u - cx)/cx, (v - cy)/cy with cx, cy from the fp32 intrinsic_M_cuda (:491);  F.grid_sample(bilinear, zeros,
 *   align_corners default False).
 *   src [N][C][H][W], dmap [HW], K [9], R [N][9], t [N][3], rays [3][HW] -> out [N][C][H][W]
 */
static inline void depth_warp_coords(const float* K, const float* R, const float* t, float rx, float ry, float rz, float d,
                                     int W, int H, float* X, float* Y, float* P, float* ix, float* iy) {
    X[0] = d * rx; X[1] = d * ry; X[2] = d * rz;
    for (int i = 0; i < 3; ++i) {
        float s = R[3 * i] * X[0];
        s = fmaf(R[3 * i + 1], X[1], s);
        s = fmaf(R[3 * i + 2], X[2], s);
        Y[i] = fmaf(t[i], 1.0f, s);
    }
    for (int i = 0; i < 3; ++i) {
        float s = K[3 * i] * Y[0];
        s = fmaf(K[3 * i + 1], Y[1], s);
        s = fmaf(K[3 * i + 2], Y[2], s);
        P[i] = s;
    }
    float u = P[0] / P[2], v = P[1] / P[2];
    float cx = K[2], cy = K[5];
    float gx = (u - cx) / cx, gy = (v - cy) / cy;
    *ix = unnormalize(gx, W, 0);
    *iy = unnormalize(gy, H, 0);
}

int oracle_warp_depth_fwd(const float* src, const float* dmap, const float* K, const float* R, const float* t,
                          const float* rays, int N, int C, int H, int W, float* out) {
    size_t hw = (size_t)H * W;
#pragma omp parallel for schedule(static)
    for (long np_ = 0; np_ < (long)N * (long)hw; ++np_) {
        int n = (int)(np_ / (long)hw);
        size_t p = (size_t)(np_ - (long)n * (long)hw);
        float X[3], Y[3], P[3], ix, iy;
        depth_warp_coords(K, R + 9 * n, t + 3 * n, rays[p], rays[hw + p], rays[2 * hw + p], dmap[p], W, H, X, Y, P, &ix, &iy);
        taps2d tp;
        bilinear_taps(ix, iy, W, H, &tp);
        for (int c = 0; c < C; ++c) {
            const float* pl = src + ((size_t)n * C + c) * hw;
            float s = 0.f;
            for (int j = 0; j < 2; ++j)
                for (int i = 0; i < 2; ++i)
                    if (tp.vx[i] && tp.vy[j]) s = s + pl[(size_t)tp.y[j] * W + tp.x[i]] * (tp.wx[i] * tp.wy[j]);
            out[((size_t)n * C + c) * hw + p] = s;
        }
    }
    return 0;
}

/* Gradient of sum(out * g_out) w.r.t. R_n and t_n (what torch autograd returns for the reference's graph: grid_sample's
 * bilinear backward w.r.t. the grid — ATen grid_sampler_2d_backward: taps outside the image contribute nothing, the
 * un-normalisation contributes W/2, H/2 — chained through the two divisions and the two matmuls).  Accumulated in double. */
int oracle_warp_depth_bwd(const float* src, const float* dmap, const float* K, const float* R, const float* t,
                          const float* rays, const float* g_out, int N, int C, int H, int W, float* g_R, float* g_t) {
    size_t hw = (size_t)H * W;
    for (int n = 0; n < N; ++n) {
        double aR[9] = {0}, at[3] = {0};
        for (size_t p = 0; p < hw; ++p) {
            float X[3], Y[3], P[3], ix, iy;
            depth_warp_coords(K, R + 9 * n, t + 3 * n, rays[p], rays[hw + p], rays[2 * hw + p], dmap[p], W, H, X, Y, P, &ix, &iy);
            taps2d tp;
            bilinear_taps(ix, iy, W, H, &tp);
            float x0 = floorf(ix), y0 = floorf(iy);
            float fx = ix - x0, fy = iy - y0;
            double gix = 0, giy = 0;
            for (int c = 0; c < C; ++c) {
                const float* pl = src + ((size_t)n * C + c) * hw;
                float g = g_out[((size_t)n * C + c) * hw + p];
                float v00 = (tp.vx[0] && tp.vy[0]) ? pl[(size_t)tp.y[0] * W + tp.x[0]] : 0.f;
                float v01 = (tp.vx[1] && tp.vy[0]) ? pl[(size_t)tp.y[0] * W + tp.x[1]] : 0.f;
                float v10 = (tp.vx[0] && tp.vy[1]) ? pl[(size_t)tp.y[1] * W + tp.x[0]] : 0.f;
                float v11 = (tp.vx[1] && tp.vy[1]) ? pl[(size_t)tp.y[1] * W + tp.x[1]] : 0.f;
                gix += (double)g * ((double)(v01 - v00) * (1.0 - fy) + (double)(v11 - v10) * fy);
                giy += (double)g * ((double)(v10 - v00) * (1.0 - fx) + (double)(v11 - v01) * fx);
            }
            double cx = K[2], cy = K[5];
            double du = gix * (W * 0.5) / cx, dv = giy * (H * 0.5) / cy;      /* d/du of ((g+1)W-1)/2 with g = (u-cx)/cx */
            double pz = P[2];
            double dP[3] = {du / pz, dv / pz, -(du * P[0] + dv * P[1]) / (pz * pz)};
            double dY[3];
            for (int i = 0; i < 3; ++i) dY[i] = K[i] * dP[0] + K[3 + i] * dP[1] + K[6 + i] * dP[2];   /* K^T dP */
            for (int i = 0; i < 3; ++i) {
                at[i] += dY[i];
                for (int j2 = 0; j2 < 3; ++j2) aR[3 * i + j2] += dY[i] * X[j2];
            }
        }
        for (int i = 0; i < 9; ++i) g_R[9 * n + i] = (float)aR[i];
        for (int i = 0; i < 3; ++i) g_t[3 * n + i] = (float)at[i];
    }
    return 0;
}

/* Check of csrc/common.hpp::div_by_const (a / c for a loop-invariant c as q = a rc, r = fma(-q, c, a), q + r rc): number of
 * finite fp32 dividends a (every `stride`-th bit pattern, |a| in [1e-30, 1e30] or 0) whose result differs from a / c. */
long oracle_div_const_mismatches(float c, long stride) {
    float rc = (float)(1.0 / (double)c);
    long bad = 0;
#pragma omp parallel for reduction(+:bad)
    for (long b = 0; b < (1L << 32); b += stride) {
        unsigned int u = (unsigned int)b;
        float a;
        __builtin_memcpy(&a, &u, 4);
        if (!isfinite(a)) continue;
        float fa = fabsf(a);
        if (fa != 0.f && (fa < 1e-30f || fa > 1e30f)) continue;
        float q = a * rc;
        float r = fmaf(-q, c, a);
        float q2 = fmaf(r, rc, q);
        if (q2 != a / c) bad++;
    }
    return bad;
}
