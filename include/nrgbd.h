/*
 * nrgbd.h — C-ABI of libnrgbd_hip.so: the MI355X (gfx950) plane-sweep depth path.
 *
 * This is the drop-in boundary for the hot path of NVlabs/neuralrgbd
 * (homography warp -> D-candidate cost volume -> D-Net -> K-Net DPV predict/update).
 * The reference has no FFI of its own: its seam is the Python operator surface of
 * code/warping/homography.py and the ATen ops composed under code/models/.  Each entry
 * point below names the reference interface (file:line, relative to /root/reference)
 * that it replaces.  The Python host (neuralrgbd_amd/) binds these with ctypes and keeps
 * the reference signatures; INTEGRATION.md shows the binding a maintainer would add.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer to contiguous fp32 unless stated otherwise;
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream); kernels are
 *     launched on the CURRENT device, are re-entrant and keep no global mutable state;
 *   - inputs are never written; outputs are fully overwritten;
 *   - return value: 0 = success, <0 = NRGBD_E_* argument error, >0 = hipError_t;
 *   - no host synchronisation, no allocation: every call is hipGraph-capturable.
 *
 * Layouts
 *   NCHW  [N][C][h][w]   what torch convolutions produce / consume
 *   NHWC  [N][h][w][Cp]  "texel" layout used by the sampling kernels; Cp = C rounded up to
 *                        a multiple of 4 so that one texel is a whole number of 16-byte
 *                        words (67 -> 68 channels = 272 B, conflict-free for ds_read_b128)
 */
#ifndef NRGBD_H
#define NRGBD_H

#ifdef __cplusplus
extern "C" {
#endif

#define NRGBD_OK             0
#define NRGBD_E_NULL        -1   /* a required pointer is NULL                      */
#define NRGBD_E_SHAPE       -2   /* a dimension is <=0 or exceeds a kernel limit    */
#define NRGBD_E_ALIGN       -3   /* Cp not a multiple of 4 / pointer not 16-B aligned */
#define NRGBD_E_ARG         -4   /* an enum / flag argument is out of range         */

#define NRGBD_DIST_L2        0   /* homography.py:81-83  img_dis_L2_pard */
#define NRGBD_DIST_L1        1   /* homography.py:85-87  img_dis_L1_pard */

#define NRGBD_MAX_D        256   /* depth candidates per volume */
#define NRGBD_MAX_V         16   /* source views per window     */

/* Library / build identification ("nrgbd_hip <interface version> (gfx950, CDNA4)"), and error text for a return code.
 * The interface version changes whenever an entry point changes its signature or disappears: bind against the version you
 * were built for.  0.4 (round 4): nrgbd_costvol_bwd takes (workspace, workspace_bytes) before `stream` (since round 3: query
 * nrgbd_costvol_bwd_workspace first); nrgbd_conv3d_wino_* and nrgbd_conv_wino_dw_bn_f32 are gone; nrgbd_upsample_bilinear_ac
 * is new.  0.5 (round 6): the three BatchNorm finalisers take (collapse_count, batches_tracked) before `stream`; nrgbd_pack_nhwc takes
 * rgb4, nrgbd_conv2d_taps_f32 takes in_stride; nrgbd_avgpool_cl, nrgbd_scatter_channels and nrgbd_conv2d_few_f32 are new. */
#define NRGBD_INTERFACE_VERSION "0.5"
const char* nrgbd_version(void);
const char* nrgbd_strerror(int code);

/*
 * nrgbd_homography_terms — the per-view constants of the plane sweep.
 * Replaces: warping/homography.py:315-317 (term1 = IntM.matmul(t_v); left factor IntM.matmul(R_v) of term2), also
 * :250-253 / :203-206 of the K-Net warps.  Summation order = the reference's CPU execution (fma chain for K R_v,
 * (p1 + p2) + p0 for K t_v) so that sampling coordinates agree with it bit for bit.
 *   K  [3][3]                 intrinsics at grid resolution
 *   R  element (v,i,j) at R[v*r_view_stride + i*r_row_stride + j]   (e.g. poses[:, :3, :3]: 16, 4)
 *   t  element (v,i)   at t[v*t_view_stride + i*t_elem_stride]      (e.g. poses[:, :3, 3]:  16, 4)
 *   KR [V][9], Kt [V][3]      outputs
 */
int nrgbd_homography_terms(const float* K, const float* R, long r_view_stride, long r_row_stride,
                           const float* t, long t_view_stride, long t_elem_stride, float* KR, float* Kt,
                           int V, void* stream);

/*
 * nrgbd_pose_inverse — inverse of the 4x4 rigid motions the PREDICT step resamples through.
 * Replaces: test_utils/test_KVNet.py:50,52 (`Src_CamPoses[ibatch, t_win_r].inverse()` / `cam_pose_next.inverse()`), the
 * argument `rel_extM` of warping/homography.py:654 resample_vol_cuda.  The reference leaves the operation order to the
 * host LAPACK; here it is fixed: Gauss-Jordan with partial pivoting on [A | I] in fp64 (every product / difference /
 * quotient rounded once), result rounded to fp32 (<= 0.5 ulp + double rounding from exact); the CPU oracle executes the
 * same sequence bit for bit.  General 4x4 (no rigidity assumed, like `.inverse()`).
 *   T      n matrices, matrix m at T + m*matrix_stride, row-major 4x4 (matrix_stride >= 16)
 *   T_inv  [n][16]
 *   singular_count   optional device int, incremented once per matrix with a zero pivot column (its output is NaN:
 *                    the reference raises on the host; a device entry point cannot, and never syncs)
 */
int nrgbd_pose_inverse(const float* T, long matrix_stride, float* T_inv, int* singular_count, int n, void* stream);

/*
 * nrgbd_pack_nhwc — feature packing for the sampling kernels.
 * Replaces: models/basic.py:254-263 (F.avg_pool2d of the RGB frames + torch.cat onto the
 * CNN features) and the implicit NCHW layout handed to homography.py:293.
 *   feat [N][Cf][h][w]            CNN features (NCHW), or [N][h][w][Cf] when feat_channels_last != 0
 *   rgb  [N][3][h*pool][w*pool]   full-resolution frames, or NULL (then only a transpose)
 *   out  [N][h][w][Cp]            out[..,c] = feat[c] (c<Cf); mean of the pool x pool RGB
 *                                 window (c = Cf..Cf+2, rgb != NULL); 0 for padding
 *   rgb4 [N][h][w][4] or NULL     channels Cf .. Cf+3 of every texel once more as a compact plane (what models/KVNET.py:147-158
 *                                 warps for the K-Net: `F.avg_pool2d(img, 4)`); needs Cp >= Cf + 4
 * Requires Cp % 4 == 0 and Cp >= Cf (+3 if rgb).
 */
int nrgbd_pack_nhwc(const float* feat, const float* rgb, float* out,
                    int N, int Cf, int h, int w, int pool, int Cp, int feat_channels_last, float* rgb4, void* stream);

/*
 * nrgbd_costvol_fwd — fused homography warp + bilinear sample + cost accumulate
 * (+ optional log-softmax over the depth axis).
 * Replaces: warping/homography.py:293-331 est_swp_volume_v4, :421-448
 * _back_warp_homo_parallel (F.grid_sample bilinear / zeros), :81-87 img_dis_L2/L1_pard,
 * and models/basic.py:299-300 (log_softmax(-costV, dim=1)) when out_logp != NULL.
 *   ref_nhwc [h][w][Cp], src_nhwc [V][h][w][Cp]   packed features (channels >= C ignored)
 *   KR [V][9]   row-major K·R_v          (the matmul of homography.py:317, left factor)
 *   Kt [V][3]   K·t_v                    (homography.py:315, term1)
 *   rays [3][h*w]                        cam_intrinsic['unit_ray_array_2D']
 *   d_candi [D]                          candidate depths (fp32 cast of the float64 array)
 *   cx, cy                               cam_intrinsic['intrinsic_M'][0,2], [1,2]
 *   sigma                                costV_sigma
 *   dist                                 NRGBD_DIST_L2 | NRGBD_DIST_L1
 *   align_corners                        0 = torch>=1.3 default (what the reference runs
 *                                        today), 1 = legacy grid_sample behaviour
 *   out_cost [D][h][w] or NULL, out_logp [D][h][w] or NULL (at least one non-NULL)
 * Views are accumulated in the order v = 0..V-1: cost += sum_c(.)/sigma.
 */
int nrgbd_costvol_fwd(const float* ref_nhwc, const float* src_nhwc,
                      const float* KR, const float* Kt, const float* rays,
                      const float* d_candi, float cx, float cy, float sigma,
                      int dist, int align_corners,
                      float* out_cost, float* out_logp,
                      int V, int C, int Cp, int D, int h, int w, void* stream);

/* The same operation with the kernel generation chosen by the caller (tests and A/B measurements):
 * 0 = automatic (what nrgbd_costvol_fwd does), 1 = direct gather (any shape), 2 = LDS-staged, lane = pixel (Cp/4 in
 * {1,2,3,4,8,9,16,17}), 3 = quad: 4 lanes per (pixel, candidate) (Cp = 68 with C > 64, or Cp = C = 64; V <= 8; a view below 2 GB with a
 * row pitch below 8 MB and h * w < 2^23: signed 24-bit address multiplies; what 0 picks for those shapes).
 * A generation that does not support the shape returns NRGBD_E_SHAPE; nothing is substituted silently.
 * `rays`: generations 1 and 3 accept ANY ray table.  Generation 3 stages the source texels a tile can touch from the images of the
 * tile's four corner pixels — exact for the reference's pinhole table (warping/View.py:16-62: rays affine in (x, y), z = 1) — and
 * re-evaluates from global memory every (16 pixels x 4 candidates) group in which a tap falls outside that prediction, so a
 * unit-norm or distortion-corrected table costs time, never correctness (tests/test_gpu_ops.py: non-affine rays vs the C oracle).
 * Generation 2 (not selected for the path's 64(+3)-channel texels) relies on the corner prediction with one texel of slack and
 * REQUIRES the pinhole table; pass generation 1 for any other table with its channel counts. */
int nrgbd_costvol_fwd_gen(const float* ref_nhwc, const float* src_nhwc,
                      const float* KR, const float* Kt, const float* rays,
                      const float* d_candi, float cx, float cy, float sigma,
                      int dist, int align_corners,
                      float* out_cost, float* out_logp,
                      int V, int C, int Cp, int D, int h, int w, int generation, void* stream);

/*
 * nrgbd_costvol_bwd — backward of nrgbd_costvol_fwd with respect to the packed features (training).
 * Replaces: what autograd records for warping/homography.py:293-331 (grid_sample backward scatter-add,
 * broadcast subtract, channel sum).  Same geometry arguments as the forward.
 *   g_cost [D][h][w]  gradient of the loss w.r.t. out_cost
 *   g_ref  [h][w][Cp], g_src [V][h][w][Cp]   overwritten
 *   workspace         device scratch of at least nrgbd_costvol_bwd_workspace() bytes, 16-byte aligned (per-slice
 *                     partial sums of the LDS scatter kernel); may be NULL when that size is 0 (grids whose
 *                     16*h*w bytes exceed the LDS budget use global atomics instead).
 * NRGBD_E_NULL / NRGBD_E_SHAPE when a needed workspace is missing / too small.
 */
int nrgbd_costvol_bwd_workspace(int V, int Cp, int D, int h, int w, size_t* bytes);
int nrgbd_costvol_bwd(const float* ref_nhwc, const float* src_nhwc,
                      const float* KR, const float* Kt, const float* rays,
                      const float* d_candi, float cx, float cy, float sigma,
                      int dist, int align_corners, const float* g_cost,
                      float* g_ref, float* g_src,
                      int V, int C, int Cp, int D, int h, int w,
                      void* workspace, size_t workspace_bytes, void* stream);

/*
 * nrgbd_bn_cl_fwd / nrgbd_bn_cl_bwd — train-mode BatchNorm (batch statistics, biased variance) with its ReLU and residual add,
 * forward and backward, on channels-last activations viewed as x [rows][C] (training path, BASELINE config 4).
 * Replaces what autograd records for models/psm_submodule.py:10-16,31-50 (convbn / BasicBlock: nn.BatchNorm2d, ReLU, `out += x`)
 * and models/basic.py:53-68,71-94 (convbn_3d / KV_NET_BASIC: nn.BatchNorm3d, ReLU, residual adds) under
 * train_utils/train_KVNet.py:152 (loss.backward()).
 *   y = act(x*scale + shift) + res        res may be NULL; relu 0/1
 *   coef [4][C]   scale = gamma*invstd, shift = beta - mean*scale, mean, invstd   (forward writes, backward reads)
 *   running_mean / running_var: both NULL, or updated in place with `momentum` (unbiased variance), as nn.BatchNorm does
 *   backward: gx [rows][C], g_gamma [C], g_beta [C]; the gradient of `res` is gy itself; coef2 [2][C] is scratch
 *   partial: scratch of nrgbd_bn_cl_workgroups(rows, C) x 2*C floats (per-workgroup partial sums, added in index order in double)
 * C % 4 == 0 and 256 % (C/4) == 0 (NRGBD_E_SHAPE otherwise).
 */
int nrgbd_bn_cl_workgroups(long rows, int C);
int nrgbd_bn_cl_fwd(const float* x, const float* res, const float* gamma, const float* beta, float eps, float momentum,
                    float* running_mean, float* running_var, int relu, float* y, float* coef, float* partial,
                    long rows, int C, void* stream);
int nrgbd_bn_cl_bwd(const float* x, const float* gy, const float* coef, int relu, float* gx, float* g_gamma,
                    float* g_beta, float* coef2, float* partial, long rows, int C, void* stream);

/*
 * nrgbd_warp_volume — plane-sweep warp of low-channel maps with the samples kept, plus
 * the K-Net input-volume assembly.
 * Replaces: warping/homography.py:234-280 warp_img_feats_v3 / :183-232 warp_img_feats_mgpu
 * (C=3, output transposed to [C][D][h][w]) and models/KVNET.py:163-166 (torch.cat of the
 * warped sources, the reference RGB repeated over D, and BV_cur - BV_predict).
 *   src: V maps of Cs channels addressed as src + v*sv + c*sc + y*sy + x*sx (elements),
 *        so both planar [V][Cs][h][w] and NHWC slices are accepted
 *   ref: one map of Cs channels with strides rc, ry, rx, or NULL
 *   bv_cur, bv_pred [D][h][w] or NULL (both or neither)
 *   out [V*Cs (+Cs if ref) (+1 if bv_cur)][D][h][w]:
 *        channel v*Cs+c   = warped source v, channel c, at depth k
 *        next Cs channels = ref[c] repeated over D
 *        last channel     = bv_cur - bv_pred
 *   channels_last: 0 = out [Ch][D][h][w] (the torch layout of KVNET.py:166),
 *                  1 = out [D][h][w][Ch] (the layout nrgbd_conv3d_3x3x3_f32 consumes)
 */
int nrgbd_warp_volume(const float* src, long sv, long sc, long sy, long sx,
                      const float* ref, long rc, long ry, long rx,
                      const float* KR, const float* Kt, const float* rays,
                      const float* d_candi, float cx, float cy, int align_corners,
                      const float* bv_cur, const float* bv_pred,
                      float* out, int V, int Cs, int D, int h, int w, int channels_last,
                      void* stream);

/*
 * nrgbd_dpv_resample — PREDICT step: rigid 3-D resample of the DPV into the next frame.
 * Replaces: warping/homography.py:654-723 resample_vol_cuda (d_candi_new=None),
 * :873-887 _set_vol_border and the .clamp(max=0,min=-1000) of
 * test_utils/test_KVNet.py:54-59.  The [1,D,h,w,3] point grid the reference builds on the
 * host and copies to the device every frame (homography.py:673-682) is generated in-kernel.
 *   dpv [D][h][w]     log-probability volume
 *   T [16]            row-major 4x4 rel_extM (DEVICE pointer: no host read, no sync)
 *   rays [3][h*w]     unit_ray_array_2D (fp32 of cam_intrinsic['unit_ray_array'])
 *   d_candi [D]
 *   tan_hh, tan_hv    tan(radians(hfov)/2), tan(radians(vfov)/2)
 *   z_half, z_radius  (z_max+z_min)/2, (z_max-z_min)/2 of fp32(d_candi)  (:689-693)
 *   pad_value         border value written on the 6 faces (log(1/D))
 *   do_clamp          clamp the result to [clamp_lo, clamp_hi]
 *   out [D][h][w]
 */
int nrgbd_dpv_resample(const float* dpv, const float* T, const float* rays,
                       const float* d_candi, float tan_hh, float tan_hv,
                       float z_half, float z_radius, float pad_value,
                       int do_clamp, float clamp_lo, float clamp_hi,
                       float* out, int D, int h, int w, void* stream);

/*
 * nrgbd_dpv_resample_to — the same resample onto a DIFFERENT set of candidate depths.
 * Replaces: warping/homography.py:654-723 resample_vol_cuda(..., d_candi_new=...) as called by the local bundle
 * adjustment driver (test_KVNet_LBA.py:414-417): the sample points are d_candi_new[k] * ray (:675-682), the depth axis is
 * normalised with z_half / z_radius of the SOURCE candidates (:686-687: float64 numpy min/max, cast to fp32 when it
 * meets the fp32 tensor), the output has D_out = len(d_candi_new) planes.
 *   dpv [D_src][h][w], d_candi_out [D_out], out [D_out][h][w]; other arguments as nrgbd_dpv_resample.
 */
int nrgbd_dpv_resample_to(const float* dpv, const float* T, const float* rays,
                          const float* d_candi_out, float tan_hh, float tan_hv,
                          float z_half, float z_radius, float pad_value,
                          int do_clamp, float clamp_lo, float clamp_hi,
                          float* out, int D_src, int D_out, int h, int w, void* stream);

/*
 * nrgbd_logsoftmax_d — log-softmax over the depth axis of scale*a (+ b).
 * Replaces: models/basic.py:299-300 (scale=-1, b=NULL) and the UPDATE step
 * models/KVNET.py:172-173 (scale=1, a = K-Net gain, b = BV_predict).
 *   a, b, out [D][n]   (n = h*w; b may be NULL; out may alias a)
 */
int nrgbd_logsoftmax_d(const float* a, const float* b, float scale, float* out,
                       int D, long n, void* stream);

/*
 * nrgbd_depth_regress — expected depth and confidence of a log-DPV.
 * Replaces: mutils/misc.py:532-548 depth_val_regression (BV_log=True; a Python loop over
 * D) and the max_d of test_utils/export_res.py:58-59.
 *   logp [D][n]; d_candi [D]; depth [n] or NULL; conf [n] or NULL (max_k logp)
 */
int nrgbd_depth_regress(const float* logp, const float* d_candi,
                        float* depth, float* conf, int D, long n, void* stream);

/*
 * nrgbd_export_depth_u16 — export epilogue of a (refined) log-DPV in one pass.
 * Replaces: test_utils/export_res.py:43-75 export_res_img — a D-slice tensor of depth values, torch.sum(exp(BV) * vol),
 * torch.max, two host round trips and `(map * scale).astype(np.uint16)` on the CPU.
 *   logp [D][n]; d_candi [D]
 *   depth [n] or NULL   sum_k exp(logp_k) d_k          conf [n] or NULL   exp(max_k logp_k)
 *   depth_u16, conf_u16 [n] or NULL   (map * scale) truncated toward zero (numpy astype), clamped to [0, 65535]
 */
int nrgbd_export_depth_u16(const float* logp, const float* d_candi, float depth_scale, float conf_scale,
                           float* depth, float* conf, unsigned short* depth_u16, unsigned short* conf_u16,
                           int D, long n, void* stream);

/*
 * nrgbd_warp_depth_fwd / _bwd — photometric warp through a per-pixel depth map and its gradient w.r.t. the poses.
 * Replaces: warping/homography.py:479-528 back_warp_th_Rt_msrc (N views) and :530-575 back_warp_th_Rt (N = 1), and the
 * autograd graph ICP/opt_pose_numerical.py:245-294 differentiates (local bundle adjustment refines R_n, t_n).
 *   src [N][C][H][W]; dmap [H][W]; K [3][3]; R [N][3][3]; t [N][3] (reference -> source n); rays [3][HW]
 *   out [N][C][H][W]          out[n,c,p] = bilinear(src[n,c], K (R_n dmap[p] ray_p + t_n) / z), zeros padding
 *   g_out [N][C][H][W]        upstream gradient
 *   partial [N][nrgbd_warp_depth_bwd_workgroups(H, W)][12]   scratch (fixed-order reduction: reproducible)
 *   g_R [N][3][3], g_t [N][3] d sum(out * g_out) / d(R_n, t_n)
 */
int nrgbd_warp_depth_fwd(const float* src, const float* dmap, const float* K, const float* R, const float* t,
                         const float* rays, float* out, int N, int C, int H, int W, void* stream);
int nrgbd_warp_depth_bwd_workgroups(int H, int W);
int nrgbd_warp_depth_bwd(const float* src, const float* dmap, const float* K, const float* R, const float* t,
                         const float* rays, const float* g_out, float* partial, float* g_R, float* g_t,
                         int N, int C, int H, int W, void* stream);

/*
 * K-Net: 3x3x3 convolution (stride 1, padding 1, no bias) on the fp32 matrix cores, with the
 * BatchNorm3d / ReLU / residual work of the reference fused around it.
 * Replaces, per layer of models/basic.py:71-94,113-132 (KV_NET_BASIC): nn.Conv3d + nn.BatchNorm3d
 * (psm_submodule.py:19-23, batch statistics) + nn.ReLU + the residual adds of :127-131.
 *
 * Activations are channels-last [D][H][W][C].  A layer reads its input as
 *       in = act(x * s + t)  [+ act(res * s' + t')]           (s,t per channel; act = ReLU or identity)
 * i.e. the normalise/affine/ReLU/add of the PREVIOUS layers is applied on the fly while the input
 * tile is loaded; `materialized` (optional) receives `in` for use as a later residual.  The raw
 * convolution output is written to y together with per-workgroup partial sums for the batch statistics:
 *   stats [nrgbd_conv3d_workgroups(D,H,W)][128]   (sum of y per channel, then sum of y^2 per channel)
 * nrgbd_bn3d_finalize reduces them (in double) to scale_shift [64][2] = (gamma*invstd, beta - mean*gamma*invstd)
 * and applies the train-mode running-statistics update (momentum, unbiased variance).
 *   x_ss / res_ss [Cin][2] or NULL (identity); x_relu / res_relu 0|1; res, materialized, stats may be NULL
 *   w_packed: nrgbd_conv3d_pack_weights(w [64][Cin][3][3][3]) -> [27*Cin*64] floats
 *   Cin in {16, 64}; Cout = 64.
 * nrgbd_conv3d_3x3x3_cout1_f32 is the last layer (Conv3d(64,1), basic.py:92-94):
 *   w_tap_major [27][64] = w[0][c][tap] transposed; y [D][H][W].
 */
int nrgbd_conv3d_workgroups(int D, int H, int W);
int nrgbd_conv3d_pack_weights(const float* w, float* w_packed, int Cin, void* stream);
int nrgbd_conv3d_3x3x3_f32(const float* x, const float* x_ss, int x_relu,
                           const float* res, const float* res_ss, int res_relu,
                           float* materialized, const float* w_packed, float* y, float* stats,
                           int D, int H, int W, int Cin, int Cout, void* stream);
int nrgbd_conv3d_3x3x3_cout1_f32(const float* x, const float* x_ss, int x_relu,
                                 const float* res, const float* res_ss, int res_relu,
                                 const float* w_tap_major, float* y,
                                 int D, int H, int W, int Cin, void* stream);
/*
 * nrgbd_conv3d_cout1_{dgrad,wgrad}_f32 — the two gradients of that last layer (training; what autograd computes for
 * nn.Conv3d(64, 1, 3, padding 1) of models/basic.py:92-94 in train_utils/train_KVNet.py:103-171).  One output channel makes every
 * direction a 27-tap stencil per voxel (memory bound); until interface 0.5 the layer ran zero-padded to 64 outputs on the 64 -> 64 kernels.
 *   gy [D][H][W] (gradient of the layer's output), x / gx [D][H][W][64] channels-last, w_tap_major [27][64] as for the forward,
 *   dw [64][27] = torch's [1][64][3][3][3] (overwritten); workspace: nrgbd_conv3d_cout1_wgrad_workspace() bytes, 16-byte aligned.
 *   gx[v][ci] = sum_tap gy[v - off(tap)] w[ci][tap];  dw[ci][tap] = sum_v gy[v - off(tap)] x[v][ci];  fixed summation orders.
 */
int nrgbd_conv3d_cout1_dgrad_f32(const float* gy, const float* w_tap_major, float* gx, int D, int H, int W, void* stream);
int nrgbd_conv3d_cout1_wgrad_workspace(int D, int H, int W, size_t* bytes);
int nrgbd_conv3d_cout1_wgrad_f32(const float* x, const float* gy, float* dw, void* workspace, size_t workspace_bytes, int D, int H, int W,
                                 void* stream);
/*
 * nrgbd_conv3d_wgrad_f32 — weight gradient of the 3x3x3 convolution (training):
 *   dW[co][ci][kd][kh][kw] = sum_voxels gy[v][co] * x[v + tap][ci]      (x zero outside the volume)
 * Replaces what autograd/MIOpen compute for nn.Conv3d.weight.grad in models/basic.py:71-94.
 *   x [D][H][W][Cin], gy [D][H][W][64] channels-last; dw [64][Cin][3][3][3] (torch layout, overwritten);
 *   partial: scratch of nrgbd_conv3d_wgrad_workgroups() * 27 * 64 * Cin floats.
 * The data gradient needs no new kernel: it is nrgbd_conv3d_3x3x3_f32 applied to gy with the weights
 * transposed (cin <-> cout) and flipped in all three tap axes.
 */
int nrgbd_conv3d_wgrad_workgroups(void);
int nrgbd_conv3d_wgrad_f32(const float* x, const float* gy, float* partial, float* dw,
                           int D, int H, int W, int Cin, void* stream);
int nrgbd_bn3d_finalize(const float* stats, int num_workgroups, long count,
                        const float* gamma, const float* beta, float eps, float momentum,
                        float* running_mean, float* running_var, float* scale_shift, unsigned int* collapse_count, long long* batches_tracked, void* stream);

/*
 * 2-D feature CNN helpers (NCHW, HW % 4 == 0, 16-B aligned planes).
 * nrgbd_bn2d_train_act — train-mode BatchNorm2d (batch statistics over N*H*W, biased variance) fused with
 * its activation and the residual add: y = act(gamma*(x-mean)/sqrt(var+eps) + beta) (+ residual).
 * Replaces: psm_submodule.py:10-16 (convbn's nn.BatchNorm2d, always batch statistics: SURVEY §0.2),
 * the nn.ReLU(inplace=True) after it (:37,:94-96) and `out += x` of BasicBlock (:47).
 *   act: 0 none, 1 ReLU; residual NULL or [N][C][HW]; y may alias x;
 *   partial: scratch of nrgbd_bn2d_partial_floats(C) floats; mean_var [C][2] or NULL (batch mean, biased var:
 *   input of the running-statistics update of the shortcut norms, psm_submodule.py:131).
 * nrgbd_avgpool8 — 8x8/stride-8 average pooling (the finest SPP window, psm_submodule.py:115; the 16/32/64
 * windows of :103-111 are pooled from its output).  x [NC][H][W] -> y [NC][H/8][W/8].
 */
int nrgbd_bn2d_partial_floats(int C);
int nrgbd_bn2d_train_act(const float* x, const float* gamma, const float* beta, float eps, int act,
                         const float* residual, float* y, float* partial, float* mean_var,
                         int N, int C, long HW, void* stream);
int nrgbd_avgpool8(const float* x, float* y, int NC, int H, int W, void* stream);
/*
 * nrgbd_avgpool_cl — k x k / stride-k average pooling of a CHANNELS-LAST map: x [N][H][W][C] -> y [N][H/k][W/k][C] (floor: a ragged
 * border is dropped, as F.avg_pool2d does).  Replaces: the four nn.AvgPool2d of models/psm_submodule.py:100-117 on the inference
 * path (window 8 on the deep map, then 2 / 4 / 8 on its result: equal windows, the mean of means is the mean).  C % 4 == 0.
 * nrgbd_scatter_channels — dst[r][pixel][coff + c] = src[c * stride_c + y * stride_y + x * stride_x] for r < n_rep images of pixel
 * stride ldy, `rep_stride` floats apart: the image features models/Refine.py:88-98 concatenates behind the candidate channels
 * (`torch.cat((dpv, feat), dim=1)`), written straight into the R-Net's channels-last concat buffers for every sample of the batch.
 */
int nrgbd_avgpool_cl(const float* x, float* y, int N, int H, int W, int C, int k, void* stream);
/*
 * nrgbd_conv2d_few_f32 — 3x3 convolution (stride 1, pad 1) + bias + optional LeakyReLU(0.01) with 1..4 OUTPUT channels, on the vector
 * ALUs.  Replaces: the three output columns beyond the 64-column groups of models/Refine.py:64-66 conv2 (conv2d_leakyRelu(D + 3, D + 3):
 * 67 = 64 + 3, 131 = 128 + 3 outputs), which otherwise cost a whole extra pass of the matrix-core kernel over the full-resolution buffer.
 *   x [N][H][W][ldx] channels-last, channels 0 .. Cin-1 read (Cin % 16 == 0 <= ldx, ldx % 4 == 0: padding channels carry zero weights)
 *   w_packed [Cin/16][9 taps = ky*3 + kx][Cout][16]  = w[co][16*block + lane][ky][kx]     (host mirror: nets.DPVUpsampleNet._rnet_packed)
 *   y [N][H][W][ldy]: output column co at y[pixel * ldy + ycoff + co]; the rest of the pixel is left alone
 * Summation per output: bias, then (block, tap, channel) ascending — one fp32 FMA chain.
 */
int nrgbd_conv2d_few_f32(const float* x, int ldx, const float* w_packed, const float* bias, int out_lrelu, float* y, int ldy,
                         int ycoff, int N, int H, int W, int Cin, int Cout, void* stream);
int nrgbd_scatter_channels(const float* src, long stride_c, long stride_y, long stride_x, int C, int H, int W, float* dst,
                           int ldy, int coff, int n_rep, long rep_stride, void* stream);
/*
 * nrgbd_bias_act_nchw — x = leaky_relu(x + bias[c], slope) in place on [N][C][HW] (HW % 4 == 0): the bias + LeakyReLU
 * tail of the R-Net's conv2d_leakyRelu / conv2dTranspose_leakyRelu blocks (models/m_submodule.py:18-27,36-45) in one
 * pass after a bias-free vendor convolution; slope = 1 is the plain bias add of Refine.py:71.
 */
int nrgbd_bias_act_nchw(float* x, const float* bias, float slope, int N, int C, long HW, void* stream);

/*
 * nrgbd_conv_wino_f32 — generation 2 of the Winograd-domain convolution (csrc/wino_pc.hip): a persistent 8-wave workgroup per
 * CU, 4 consumer waves that only issue MFMAs and 4 producer waves that load / normalise / transform the operand two stages
 * ahead.  One entry for
 *   kd = 3: the K-Net's 3x3x3 layers (models/basic.py:71-94), x [N = depth][H][W][Cin], dilation 1;
 *   kd = 1: the 3x3 stride-1 layers of the feature CNN (models/psm_submodule.py:10-16,31-50,100-134), x [N images][H][W][Cin],
 *           padding = dilation in {1, 2}.
 * Cin % 16 == 0, Cout % 64 == 0; same fused prologue (x_ss / x_relu / res / res_ss / res_relu / materialized) and statistics
 * epilogue as nrgbd_conv3d_3x3x3_f32 / nrgbd_conv2d_3x3_f32.
 * Cout = 32 (kd = 1, dilation 1: the trunk's half-resolution 32 -> 32 layers, psm_submodule.py:90-103): the kernel's HALF form —
 *   w_wino is the 64-column stream of the layer's weights with the upper 32 columns zero, stats has TWO rows per tile
 *   ([2*32][2 * nrgbd_conv_wino_tiles]: one per (tile, 16-tile row block)); nrgbd_conv_wino_rnet_ex_f32 takes Cout = 32 the same way.
 *   w_wino: [Cout/64][stage = cb*kd + depth tap][16 transform points][4 waves][64 lanes][4] floats, U = G g G^T over (ky, kx)
 *           (host mirror: neuralrgbd_amd/ops.py::conv_wino_pack)
 *   stats  [2*Cout][nrgbd_conv_wino_tiles(N,H,W,dilation)] (COLUMN-major: a channel's per-tile partial sums, then its partial
 *           sums of squares, each one contiguous run) for nrgbd_bn_finalize_cm, or NULL
 * nrgbd_bn_finalize_cm — the BatchNorm finaliser (same contract as nrgbd_bn_finalize: models/basic.py:13-51 BatchNorm2d/3d with
 *   batch statistics, running-statistics side effect) for column-major partials [2C][rows]
 */
int nrgbd_conv_wino_tiles(int N, int H, int W, int dilation);
/* R-Net form of the same kernel (models/m_submodule.py:18-27 conv2d_leakyRelu where Cin % 16 == 0 (>= 32; an odd number of 16-channel stages has its own instantiation) and Cout % 64 == 0:
 * Refine.py:51-56 conv0 / conv0_1): y = leaky_relu(conv3x3(x) + bias, 0.01 if out_lrelu), x [N][H][W][Cin] -> y [N][H][W][Cout] */
int nrgbd_conv_wino_rnet_f32(const float* x, const float* w_wino, const float* bias, int out_lrelu, float* y,
                             int N, int H, int W, int Cin, int Cout, void* stream);
/* The same with the R-Net's generalised output addressing (as nrgbd_conv2d_rnet_f32): output column c of a pixel goes to
 * y[pixel * ldy + ycoff + c] for c < cout_valid (ldy = 0: Cout, cout_valid = 0: Cout) — the half- and full-resolution layers
 * (Refine.py:57-70: 96 -> 96 and 67 -> 67 / 64, widths padded to Cin % 16 == 0, Cout % 64 == 0 with zero weights) write into
 * 96-wide concat buffers whose padding channels stay zero. */
int nrgbd_conv_wino_rnet_ex_f32(const float* x, const float* w_wino, const float* bias, int out_lrelu, float* y,
                                int N, int H, int W, int Cin, int Cout, int ldy, int ycoff, int cout_valid, void* stream);
/* w [Cout][Cin][kd][3][3] (torch layout) -> w_wino [Cout*Cin*kd*16] floats in the order described above (on the device).
 * transposed = 1: the DATA-GRADIENT weights of w [Cin][Cout][kd][3][3] (roles of the channel axes swapped, every tap flipped),
 * i.e. what the same kernel needs to turn dL/dy into dL/dx — without materialising w.transpose(0,1).flip(...) first.
 * transposed = 2: BOTH streams of the layer w [Cout][Cin][kd][3][3] in one launch (training re-packs them every iteration):
 * the forward stream at w_wino, the data-gradient stream right behind it (w_wino holds 2 * Cout*Cin*kd*16 floats; Cin % 64 too). */
int nrgbd_conv_wino_pack(const float* w, float* w_wino, int Cin, int Cout, int kd, int transposed, void* stream);
int nrgbd_bn_finalize_cm(const float* stats, int rows, int C, long count, const float* gamma, const float* beta, float eps,
                         float momentum, float* running_mean, float* running_var, float* scale_shift, unsigned int* collapse_count,
                         long long* batches_tracked, void* stream);
int nrgbd_conv_wino_f32(const float* x, const float* x_ss, int x_relu, const float* res, const float* res_ss,
                        int res_relu, float* materialized, const float* w_wino, float* y, float* stats,
                        int N, int H, int W, int Cin, int Cout, int kd, int dilation, void* stream);

/*
 * nrgbd_conv_wino_dw_f32 — generation 3 of the K-Net's 3x3x3 layers (csrc/wino_dw.hip; models/basic.py:71-94): Winograd in all
 * three dimensions, F(2x2, 3x3) in the plane and F(2, 3) along the depth axis: 8 fp32 multiplies per output voxel (generation
 * 2: 12, direct: 27), exact-algorithm fp32.  Same contract as nrgbd_conv_wino_f32 with kd = 3 (x [N = depth][H][W][Cin], fused
 * prologue, column-major statistics [2*Cout][nrgbd_conv_wino_tiles(N,H,W,1)] for nrgbd_bn_finalize_cm), restricted to N even
 * and whole 8x16 tiles (H % 8 == 0, W % 16 == 0: every grid of the path); other shapes return NRGBD_E_SHAPE and belong to
 * nrgbd_conv_wino_f32 — nothing is substituted silently.
 *   w_wino: [Cout/64][stage = t*(Cin/16) + cb][16 transform points][4 waves][64 lanes][4] floats,
 *           U_t = sum_kd Gd[t][kd] (G g_kd G^T), Gd = G = [1 0 0; .5 .5 .5; .5 -.5 .5; 0 0 1]  (nrgbd_conv_wino_dw_pack;
 *           transposed as in nrgbd_conv_wino_pack)
 */
int nrgbd_conv_wino_dw_pack(const float* w, float* w_wino, int Cin, int Cout, int transposed, void* stream);
int nrgbd_conv_wino_dw_f32(const float* x, const float* x_ss, int x_relu, const float* res, const float* res_ss,
                           int res_relu, float* materialized, const float* w_wino, float* y, float* stats,
                           int N, int H, int W, int Cin, int Cout, void* stream);
int nrgbd_conv_wino_dw_workgroups(int N, int H, int W, int Cout);
/*
 * nrgbd_conv_wino_dw4_* — the same 3x3x3 convolution (models/basic.py:71-94: the K-Net's ten 64 -> 64 layers) with F(4, 3) along the
 * depth axis (interpolation points 0, +-1/2, +-3/2, inf) on top of the in-plane F(2x2, 3x3): 6 multiplies per output voxel instead of 8;
 * N % 4 == 0 (quadruples of output slices), whole 8x16 tiles.  Numerically 1.24x as far from float64 as the direct convolution on the
 * whole K-Net (oracle/wino_d4_eval.py; nrgbd_conv_wino_dw_f32: 0.92x).  Input forms: x as it is (x_ss NULL, x_relu 0), act(x * s + t),
 * or — x_unit = 2^-k > 0, x_ss given, x_relu 1 — relu(x * s + t) as a clamped FMA with w_wino packed from 2^k * w (see
 * nrgbd_conv_wino_dw_unit_f32).  `workspace`: nrgbd_conv_wino_dw4_workspace() bytes of device scratch, 16-byte aligned, private to the
 * launch (one third of the depth fold's live values per workgroup goes through it: they do not fit the LDS).
 *   w_wino: [Cout/64][stage = p * Cin/16 + cb][16 points][4 waves][64 lanes][4], phases p in execution order t = 1, 2, 3, 4, 0, 5
 *   stats  [2*Cout][nrgbd_conv_wino_tiles(N,H,W,1)] column-major partials for nrgbd_bn_finalize_cm, or NULL
 *   pack: transposed = 1 packs the data-gradient stream (w read as [Cin][Cout][3][3][3], taps flipped), 2 both streams in one launch
 *         (forward, then data gradient; Cin % 64 == 0) — as nrgbd_conv_wino_dw_pack
 */
int nrgbd_conv_wino_dw4_pack(const float* w, float* w_wino, int Cin, int Cout, int transposed, void* stream);
int nrgbd_conv_wino_dw4_workspace(int N, int H, int W, int Cout, size_t* bytes);
int nrgbd_conv_wino_dw4_f32(const float* x, const float* x_ss, int x_relu, float x_unit, const float* w_wino, float* y, float* stats,
                            void* workspace, size_t workspace_bytes, int N, int H, int W, int Cin, int Cout, void* stream);
/*
 * nrgbd_conv_wino_dw_unit_f32 — nrgbd_conv_wino_dw_f32's plain form for an input relu(x * scale + shift) (no residual, no
 * materialise) with the ReLU taken by the producers' FMA itself (its [0, 1] clamp) instead of one v_max_f32 per element:
 *   x_unit = 2^-k: the kernel multiplies (scale, shift) by it; the caller guarantees |x * scale + shift| < 2^k everywhere (for a
 *   BatchNorm with batch statistics over n values: |gamma| sqrt(n) + |beta| bounds it, whatever the data) and packs the weight
 *   stream from 2^k * w.  Powers of two commute with every rounding on the way, so y and stats have the bits of the plain form.
 * Same layer as nrgbd_conv_wino_dw_f32 (models/basic.py:53-68,71-94: conv3d + BatchNorm3d + ReLU feeding the next conv3d).
 */
int nrgbd_conv_wino_dw_unit_f32(const float* x, const float* x_ss, float x_unit, const float* w_wino, float* y, float* stats,
                                int N, int H, int W, int Cin, int Cout, void* stream);

/*
 * R-Net (DPV up-sampler) on the same matrix-core kernel.  Replaces, per layer of models/Refine.py:51-107:
 *   m_submodule.conv2d_leakyRelu (nn.Conv2d 3x3 + bias + LeakyReLU 0.01, :18-27)      mode 0
 *   m_submodule.conv2dTranspose_leakyRelu (nn.ConvTranspose2d k4 s2 p1 + bias + LeakyReLU, :37-45)
 *                                             as four sub-pixel 2x2-tap launches (pa, pb in {0,1})   mode 1
 *   conv2_2 + F.log_softmax(dim=1) (:99-105)                                          mode 2
 *   the four phases of a transposed convolution in ONE launch (blockIdx.y = phase; w_packed = the four phases' packed
 *   weights back to back in the order (pa, pb) = (0,0), (0,1), (1,0), (1,1); pa / pb ignored)          mode 3
 *   the torch.cat of :88,93,98: a layer writes its cout_valid channels at channel offset ycoff of a wider
 *   channels-last pixel (pixel stride ldy floats), i.e. straight into the next layer's concat buffer.
 * x [N][H][W][Cin] channels-last, Cin % 16 == 0 (zero-padded channels carry zero weights);
 * w_packed: nrgbd_conv_pack_weights of w [Cout][Cin][taps], taps = 9 (modes 0, 2) or 4 (mode 1: the phase's 2x2 taps in
 *   row-major order of the input neighbourhood {y-1+pa, y+pa} x {x-1+pb, x+pb}); Cout in {64, 96, 128} (mode 0), 64 (mode 1),
 *   {64, 128} (mode 2); wider layers are launched as several output-column slices (ycoff);
 * bias [Cout] (padded) or NULL.  mode 0: y[pix*ldy + ycoff + c], c < cout_valid.  mode 1: the same at output pixel
 *   (2y+pa, 2x+pb) of a [N][2H][2W] tensor.  mode 2: y = planar [N][Cout][H][W] log-probabilities (ldy.. ignored).
 * nrgbd_rnet_pack: out [P][D+Cf] = (exp(dpv_log [D][P]), feat) — the R-Net's first concat with torch.exp fused
 *   (models/KVNET.py:128,176 + Refine.py:88); feat is [P][Cf] (feat_planar = 0) or [Cf][P].
 */
int nrgbd_conv2d_rnet_f32(const float* x, const float* w_packed, const float* bias, int out_lrelu, float* y,
                          int ldy, int ycoff, int cout_valid, int mode, int pa, int pb,
                          int N, int H, int W, int Cin, int Cout, void* stream);
int nrgbd_rnet_pack(const float* dpv_log, const float* feat, int feat_planar, float* out, int D, int Cf, long P,
                    void* stream);

/*
 * Feature CNN (and R-Net form): 3x3 convolution, stride 1, padding = dilation, on the fp32 matrix cores with the
 * BatchNorm2d / ReLU / residual work fused around it — the 2-D sibling of nrgbd_conv3d_3x3x3_f32.
 * Replaces, per trunk layer of models/psm_submodule.py:90-167 (feature_extraction; convbn :10-16, BasicBlock :31-50):
 * nn.Conv2d(bias=False) + nn.BatchNorm2d (batch statistics) + nn.ReLU + `out += x`; with (bias, out_lrelu) it is the
 * conv2d_leakyRelu of models/m_submodule.py:18-27 used by the R-Net (Refine.py:36-71).
 *
 * Activations are channels-last [N][H][W][C].  A layer reads its input as
 *       in = act(x * s + t)  [+ act(res * s' + t')]          (s,t per channel; act = ReLU or identity)
 * `materialized` (optional) receives `in`.  y = conv(in) (+ bias, LeakyReLU 0.01 if out_lrelu) and
 *   stats [nrgbd_conv2d_workgroups(N,H,W)][2*Cout] = per-workgroup sum of y, then sum of y^2, per channel.
 *   w_packed: nrgbd_conv_pack_weights(w [Cout][Cin][3][3], taps = 9) -> [9*Cin*Cout] floats
 *   Cin % 16 == 0; (Cout, dilation) in {(32,1), (64,1), (96,1), (128,1), (128,2)}; N*H*W*Cin < 2^32.
 * nrgbd_bn_finalize: partials [num_workgroups][2*C] -> scale_shift [C][2] = (gamma*invstd, beta - mean*gamma*invstd),
 *   reduced in double; running_mean / running_var (both or neither) get the train-mode update.
 *   VARIANCE COLLAPSE (all three finalisers): the variance is E[y^2] - mean^2 over fp32 per-tile partials; a channel whose computed
 *   variance is below 1e-5 mean^2 (std / |mean| < 3.2e-3: no correct digit left; the reference's two-pass statistics would still
 *   normalise it) gets scale = shift = NaN AND is counted into *collapse_count (device word, may be NULL; atomicAdd of 1 per
 *   channel) — the kernels' ReLU maps NaN to 0, so the NaN alone could vanish again; the host mirror raises on a non-zero word.
 *   batches_tracked (device int64, may be NULL): nn.BatchNorm's `num_batches_tracked += 1` side effect, done by the finaliser.
 * nrgbd_nhwc_stats: the same partials for a channels-last tensor produced elsewhere (the stride-2 / 1x1 layers that
 *   stay on the vendor library): x [P][C], stats [nrgbd_nhwc_stats_workgroups(P)][2*C]; C in {32, 64, 128}.
 * nrgbd_nhwc_act: y[p*ldy + c] = act(x*s+t) [+ act(res*s'+t')] — the loader's prologue as a stand-alone pass, for
 *   consumers that are not the conv kernel (pooling, the SPP concat of psm_submodule.py:160-163, the 1x1 head).
 */
/*
 * nrgbd_conv2d_taps_f32 — the trunk's remaining convolution forms on the same kernel (round 2: no vendor convolution is left in
 * the feature CNN), with the prologue (x_ss, x_relu) and the statistics epilogue of nrgbd_conv2d_3x3_f32:
 *   taps = 1: 1x1 convolution (psm_submodule.py:40-43,63-66 shortcut of layer2 / layer3, :103-117 SPP branch convs, :122 head);
 *             w_packed = nrgbd_conv_pack_weights(w [Cout][Cin][1], taps = 1); Cout in {32, 64, 128}; in_stride = s > 1: the
 *             convolution's own stride — x is [N][H*s][W*s][Cin], output pixel (y, x) reads input pixel (y*s, x*s) (no gather pass)
 *   taps = 4: the 2x2 window {y-1, y} x {x-1, x} — a stride-2, pad-1 3x3 convolution (psm_submodule.py:90 firstconv, :120
 *             layer2's first conv) on the space-to-depth image of its input (nrgbd_space_to_depth2), weights re-indexed by the
 *             host mirror (neuralrgbd_amd/ops.py::conv_s2_pack); Cout in {32, 64}
 * nrgbd_space_to_depth2: y [N][H/2][W/2][Cp] with y[.][(py*2+px)*C + c] = x[2y+py][2x+px][c], channels >= 4C zero;
 *   x is [N][C][H][W] (nchw = 1: the input image) or [N][H][W][C]; H, W even, Cp >= 4C.
 */
int nrgbd_conv2d_taps_f32(const float* x, const float* x_ss, int x_relu, const float* w_packed, float* y, float* stats,
                          int N, int H, int W, int Cin, int Cout, int taps, int in_stride, void* stream);
int nrgbd_space_to_depth2(const float* x, int nchw, float* y, int N, int C, int H, int W, int Cp, void* stream);
/*
 * nrgbd_conv2d_wgrad_f32 — weight gradient of a 3x3 stride-1 convolution (padding = dilation in {1, 2}) on channels-last
 * activations, on the fp32 matrix cores (training: train_utils/train_KVNet.py:103-153 back-propagating through
 * models/psm_submodule.py:10-16,31-50 and models/Refine.py:51-107):
 *   dw [Cout][Cin][3][3] = sum over (n, y, x) of gy[n][y][x][co] * x[n][y+(ky-1)d][x+(kx-1)d][ci]
 * x [N][H][W][Cin], gy [N][H][W][Cout]; Cin % 16 == 0, Cout % 16 == 0.
 * partial: workspace of ceil(Cout/64) * ceil(Cin/64) * nrgbd_conv2d_wgrad_workgroups(N,H,W,Cin,Cout) * 9*64*64 floats (per-workgroup
 * partial sums, reduced in a fixed order: bitwise reproducible).  The data gradient of the same convolution is the forward
 * kernel on gy with the transposed, flipped weights.
 */
int nrgbd_conv2d_wgrad_workgroups(int N, int H, int W, int Cin, int Cout);
int nrgbd_conv2d_wgrad_f32(const float* x, const float* gy, float* partial, float* dw, int N, int H, int W, int Cin, int Cout,
                           int dilation, void* stream);
int nrgbd_conv2d_workgroups(int N, int H, int W);
int nrgbd_conv_pack_weights(const float* w, float* w_packed, int Cin, int Cout, int taps, void* stream);
int nrgbd_conv2d_3x3_f32(const float* x, const float* x_ss, int x_relu,
                         const float* res, const float* res_ss, int res_relu,
                         float* materialized, const float* w_packed, const float* bias, int out_lrelu,
                         float* y, float* stats, int N, int H, int W, int Cin, int Cout, int dilation,
                         void* stream);
int nrgbd_bn_finalize(const float* stats, int num_workgroups, int C, long count,
                      const float* gamma, const float* beta, float eps, float momentum,
                      float* running_mean, float* running_var, float* scale_shift, unsigned int* collapse_count, long long* batches_tracked, void* stream);
/*
 * nrgbd_spp_concat — the tail of the feature CNN's spatial-pyramid pooling in one channels-last pass.
 * Replaces: models/psm_submodule.py:149-161 — for the four branches nn.ReLU after convbn (:100-117 branch1..4), F.upsample(
 * bilinear, align_corners=True) (:153-158), and the torch.cat of (output_raw, output_skip, branch4, branch3, branch2, branch1)
 * (:160).  out[pixel] = [ quarter (Cq) | deep (Cd) | up(relu(bz0 * s + t)) | up(.. bz1) | up(.. bz2) | up(.. bz3) ].
 *   quarter [N][h][w][Cq], deep [N][h][w][Cd]; branch i: raw 1x1-conv output bz_i [N][bh_i][bw_i][Cb] and the (scale, shift)
 *   [Cb][2] of its BatchNorm (nrgbd_bn_finalize); out [N][h][w][Cq + Cd + 4 Cb]; all channel counts % 4 == 0,
 *   Cq + Cd + 4 Cb <= 512 (a workgroup = four pixels' 16-byte words: 320 channels in models/psm_submodule.py), h, N <= 65535.
 * Interpolation arithmetic = ATen upsample_bilinear2d (fp32 scale (in-1)/(out-1), lambda clamped to [0,1]).
 */
int nrgbd_spp_concat(const float* quarter, int Cq, const float* deep, int Cd,
                     const float* bz0, const float* bss0, int bh0, int bw0, const float* bz1, const float* bss1, int bh1, int bw1,
                     const float* bz2, const float* bss2, int bh2, int bw2, const float* bz3, const float* bss3, int bh3, int bw3,
                     int Cb, float* out, int N, int h, int w, void* stream);
/*
 * nrgbd_logsoftmax_rows — log_softmax over the channels of channels-last rows x [rows][C] (y may be x), C in {64, 128}.
 * Replaces: F.log_softmax(conv2_2_out, dim=1) of models/Refine.py:104 once the last R-Net convolution runs on the Winograd kernel
 * (nrgbd_conv_wino_rnet_ex_f32), whose pixels are channels-last; the refined DPV is then an [N, D, H, W] VIEW of that memory.
 */
int nrgbd_logsoftmax_rows(const float* x, float* y, long rows, int C, void* stream);
/*
 * nrgbd_logsoftmax_d_bwd / nrgbd_logsoftmax_rows_bwd — backward of nrgbd_logsoftmax_d / nrgbd_logsoftmax_rows (training path):
 * with out = log_softmax(z), g_z = g - exp(out) * sum_k g (the _d form returns scale * g_z = the gradient of its operand a;
 * scale = 1 gives the gradient of b).  Replaces: autograd through torch.log_softmax at models/basic.py:299-300,
 * models/KVNET.py:172-173 and models/Refine.py:104 (ATen's SpatialSoftMaxBackward).
 *   _d:    logp, g, gz [D][n];   _rows: y, g, gx [rows][C], C in {64, 128}
 */
int nrgbd_logsoftmax_d_bwd(const float* logp, const float* g, float scale, float* gz, int D, long n, void* stream);
int nrgbd_logsoftmax_rows_bwd(const float* y, const float* g, float* gx, long rows, int C, void* stream);
/*
 * nrgbd_nll_fwd / nrgbd_nll_bwd — F.nll_loss(logp [1, D, h, w], target [1, h, w], ignore_index) with mean reduction and its
 * backward.  Replaces: the four loss terms of train_utils/train_KVNet.py:103-120 (ignore_index = 0: pixels without ground truth).
 *   logp: planar [D][n] (channels_last = 0) or channels-last [n][D] (= 1: the refined DPV as the R-Net writes it); target [n] int64.
 *   fwd:  partial [nrgbd_nll_workgroups(n)][2] scratch; out[0] = mean over the counted pixels (NaN when there is none, as ATen),
 *         out[1] = their number.  A target outside [0, D) is not counted (ATen asserts).  Fixed summation order.
 *   bwd:  g_out [1] (device), stat = fwd's out; g_logp in logp's layout, every element written (zeros off the target).
 */
int nrgbd_nll_workgroups(long n);
int nrgbd_nll_fwd(const float* logp, const long long* target, long ignore_index, int D, long n, int channels_last,
                  float* partial, float* out, void* stream);
int nrgbd_nll_bwd(const long long* target, long ignore_index, const float* g_out, const float* stat, float* g_logp, int D,
                  long n, int channels_last, void* stream);
/*
 * nrgbd_adam_step — torch.optim.Adam's update (no amsgrad) of a LIST of fp32 tensors, 48 tensors per launch with their pointers passed
 * by value (nothing is uploaded: the launches can be captured into a hipGraph).  Replaces: optimizer_KV.step() of
 * train_utils/train_KVNet.py:153 for the optim.Adam of train_KVNet.py:228-232 (ATen: ~50 _foreach_ launches per step).
 *   params / grads / exp_avg / exp_avg_sq / steps: HOST arrays of `ntensors` DEVICE pointers; numel: host array of element counts.
 *   steps[i] points to tensor i's step counter (a device float: completed steps), read as t = *steps[i] + 1 and then incremented.
 *   Arithmetic = torch.optim.adam._single_tensor_adam, fp32, scalar factors formed in double:
 *     g' = (maximize ? -g : g) + weight_decay * p;  m += (g' - m)(1 - beta1);  v = v beta2 + (1 - beta2) g'^2;
 *     p -= lr / (1 - beta1^t) * m / (sqrt(v) / sqrt(1 - beta2^t) + eps)
 */
int nrgbd_adam_step(float* const* params, const float* const* grads, float* const* exp_avg, float* const* exp_avg_sq,
                    float* const* steps, const long* numel, int ntensors, double lr, double beta1, double beta2, double eps,
                    double weight_decay, int maximize, void* stream);
/*
 * nrgbd_bias_lrelu_cl_fwd / _bwd — y = leaky_relu(x + bias[c], slope) on channels-last rows [rows][C] and its backward
 * (training path of the R-Net).  Replaces: the bias add + nn.LeakyReLU of m_submodule.conv2d_leakyRelu /
 * conv2dTranspose_leakyRelu (models/m_submodule.py:18-27,36-45; slope = 1: the bias of Refine.py:71) and, in backward, ATen's
 * leaky_relu_backward + the bias gradient's sum.  backward: gx = gy * (y > 0 ? 1 : slope) (y = the forward's OUTPUT),
 * g_bias[c] = sum over rows of gx; partial = scratch of nrgbd_bias_lrelu_cl_workgroups(rows, C) x C floats (added in index
 * order, in double).  C % 4 == 0, C <= 1024.
 */
int nrgbd_bias_lrelu_cl_workgroups(long rows, int C);
int nrgbd_bias_lrelu_cl_fwd(const float* x, const float* bias, float slope, float* y, long rows, int C, void* stream);
int nrgbd_bias_lrelu_cl_bwd(const float* y, const float* gy, float slope, float* gx, float* g_bias, float* partial,
                            long rows, int C, void* stream);
/*
 * nrgbd_upsample_bilinear_ac — bilinear up-sampling with align_corners = True of a channels-last map and its exact adjoint
 * (training path of the SPP branches).
 * Replaces: F.upsample(mode='bilinear') of models/psm_submodule.py:153-158 and its autograd backward (ATen
 * upsample_bilinear2d_backward: an atomic scatter).  backward = 0: x [N][bh][bw][C] -> y [N][H][W][C];
 * backward = 1: x = gradient of the output [N][H][W][C] -> y = gradient of the input [N][bh][bw][C], every input element summed
 * by one thread in a fixed order with the forward's own fp32 weights.  C % 4 == 0.
 */
int nrgbd_upsample_bilinear_ac(const float* x, float* y, int N, int bh, int bw, int H, int W, int C, int backward,
                               void* stream);
int nrgbd_nhwc_stats_workgroups(long P);
int nrgbd_nhwc_stats(const float* x, long P, int C, float* stats, void* stream);
int nrgbd_nhwc_act(const float* x, const float* x_ss, int x_relu, const float* res, const float* res_ss,
                   int res_relu, float* y, long P, int C, int ldy, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* NRGBD_H */
