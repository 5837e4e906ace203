"""Typed torch-tensor wrappers over the C-ABI of libnrgbd_hip.so (include/nrgbd.h).

PyTorch is used here only for device memory and the current HIP stream; every function below
launches hand-written gfx950 kernels.  Inputs must be CUDA fp32 tensors — there is no CPU
path: a CPU tensor raises.
"""
import ctypes

import torch

from . import _lib

DIST = {"L2": 0, "L1": 1}
GENERATION = {None: 0, "auto": 0, "gather": 1, "lds": 2, "quad": 3}   # kernel generations of nrgbd_costvol_fwd_gen


def _need(t, name, shape=None, strided=False):
    if not isinstance(t, torch.Tensor):
        raise TypeError("%s must be a torch.Tensor" % name)
    if not t.is_cuda:
        raise _lib.NrgbdError("%s is on %s: the plane-sweep path runs on the GPU only "
                              "(no CPU fallback)" % (name, t.device))
    if t.dtype != torch.float32:
        raise TypeError("%s must be float32, got %s" % (name, t.dtype))
    if shape is not None and tuple(t.shape) != tuple(shape):
        raise ValueError("%s has shape %s, expected %s" % (name, tuple(t.shape), tuple(shape)))
    return t if (strided or t.is_contiguous()) else t.contiguous()


def _p(t):
    return ctypes.c_void_p(0 if t is None else t.data_ptr())


def _stream(t):
    return ctypes.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


def version():
    return _lib.load().nrgbd_version().decode()


def padded_channels(C):
    return (C + 3) // 4 * 4


def homography_terms(K, R, t):
    """K [3,3], R [V,3,3] (any strides with unit column stride), t [V,3] (any strides) -> (KR [V,3,3], Kt [V,3])
    in the reference's CPU summation order (nrgbd_homography_terms)."""
    K = _need(K, "K", (3, 3))
    R = _need(R, "R", strided=True)
    t = _need(t, "t", strided=True)
    V = R.shape[0]
    if tuple(R.shape) != (V, 3, 3) or tuple(t.shape) != (V, 3):
        raise ValueError("homography_terms: R %s / t %s, expected [V,3,3] / [V,3]" % (tuple(R.shape), tuple(t.shape)))
    if R.stride(2) != 1:
        R = R.contiguous()
    KR = torch.empty((V, 3, 3), dtype=torch.float32, device=K.device)
    Kt = torch.empty((V, 3), dtype=torch.float32, device=K.device)
    with torch.cuda.device(K.device):
        rc = _lib.load().nrgbd_homography_terms(_p(K), _p(R), R.stride(0), R.stride(1), _p(t), t.stride(0), t.stride(1),
                                                _p(KR), _p(Kt), V, _stream(K))
    _lib.check(rc, "nrgbd_homography_terms")
    return KR, Kt


def pose_inverse(T, singular_count=None):
    """T [...,4,4] fp32 on the GPU -> its inverse, same shape (nrgbd_pose_inverse: fp64 Gauss-Jordan in a fixed operation
    order, rounded to fp32; replaces `.inverse()` of test_utils/test_KVNet.py:50).  Capture-safe: no solver workspace,
    no host sync.  `singular_count`: optional int32 device scalar that counts singular inputs (their output is NaN)."""
    T = _need(T, "T")
    if T.dim() < 2 or tuple(T.shape[-2:]) != (4, 4):
        raise ValueError("pose_inverse: T %s, expected [...,4,4]" % (tuple(T.shape),))
    n = T.numel() // 16
    out = torch.empty_like(T)
    if singular_count is not None and (singular_count.dtype != torch.int32 or not singular_count.is_cuda):
        raise TypeError("singular_count must be an int32 CUDA tensor")
    with torch.cuda.device(T.device):
        rc = _lib.load().nrgbd_pose_inverse(_p(T), 16, _p(out), _p(singular_count), n, _stream(T))
    _lib.check(rc, "nrgbd_pose_inverse")
    return out


def pack_nhwc(feat, rgb=None, Cp=None, channels_last=False, want_rgb4=False):
    """feat [N,Cf,h,w] (or [N,h,w,Cf] if channels_last) (+ rgb [N,3,h*pool,w*pool]) -> texels [N,h,w,Cp] (nrgbd_pack_nhwc).
    want_rgb4: -> (texels, rgb4 [N,h,w,4]): the pooled-RGB word of every texel once more as a compact plane (the K-Net's warp)."""
    feat = _need(feat, "feat")
    if channels_last:
        N, h, w, Cf = feat.shape
    else:
        N, Cf, h, w = feat.shape
    pool = 1
    if rgb is not None:
        rgb = _need(rgb, "rgb")
        if rgb.shape[0] != N or rgb.shape[1] != 3 or rgb.shape[2] % h or rgb.shape[3] % w:
            raise ValueError("rgb %s does not pool onto feat %s" % (tuple(rgb.shape), tuple(feat.shape)))
        pool = rgb.shape[3] // w
        if rgb.shape[2] // h != pool:
            raise ValueError("anisotropic pooling is not supported")
    if Cp is None:
        Cp = padded_channels(Cf + (3 if rgb is not None else 0))
    out = torch.empty((N, h, w, Cp), dtype=torch.float32, device=feat.device)
    rgb4 = torch.empty((N, h, w, 4), dtype=torch.float32, device=feat.device) if want_rgb4 else None
    with torch.cuda.device(feat.device):
        rc = _lib.load().nrgbd_pack_nhwc(_p(feat), _p(rgb), _p(out), N, Cf, h, w, pool, Cp, int(bool(channels_last)),
                                          _p(rgb4), _stream(feat))
    _lib.check(rc, "nrgbd_pack_nhwc")
    return (out, rgb4) if want_rgb4 else out


def costvol(ref_nhwc, src_nhwc, KR, Kt, rays, d_candi, cx, cy, sigma, C, dist="L2",
            align_corners=False, want_cost=True, want_logp=False, generation=None):
    """Fused warp + cost volume (+ log-softmax).  Returns (cost [D,h,w] | None, logp [D,h,w] | None).
    `generation` (None = automatic | "gather" | "lds" | "quad") pins the kernel generation for tests / A-B timing
    (automatic = "quad" for the path's 64(+3)-channel texel; "quad4" = the same kernel with its per-view accumulators in LDS:
    bit-identical, every output written once, measured slower — profiles/r3_costvol_gen4.txt)."""
    src_nhwc = _need(src_nhwc, "src_nhwc")
    V, h, w, Cp = src_nhwc.shape
    ref_nhwc = _need(ref_nhwc, "ref_nhwc", (h, w, Cp))
    KR = _need(KR, "KR").reshape(V, 9)
    Kt = _need(Kt, "Kt", (V, 3))
    rays = _need(rays, "rays", (3, h * w))
    d_candi = _need(d_candi, "d_candi")
    D = d_candi.numel()
    dev = src_nhwc.device
    cost = torch.empty((D, h, w), dtype=torch.float32, device=dev) if want_cost else None
    logp = torch.empty((D, h, w), dtype=torch.float32, device=dev) if want_logp else None
    with torch.cuda.device(dev):
        rc = _lib.load().nrgbd_costvol_fwd_gen(_p(ref_nhwc), _p(src_nhwc), _p(KR), _p(Kt), _p(rays),
                                               _p(d_candi), float(cx), float(cy), float(sigma),
                                               DIST[dist], int(bool(align_corners)), _p(cost), _p(logp),
                                               V, int(C), Cp, D, h, w, GENERATION[generation], _stream(src_nhwc))
    _lib.check(rc, "nrgbd_costvol_fwd_gen")
    return cost, logp


def costvol_bwd(ref_nhwc, src_nhwc, KR, Kt, rays, d_candi, cx, cy, sigma, C, g_cost, dist="L2", align_corners=False):
    """Gradient of the cost volume w.r.t. the packed features: (g_ref [h,w,Cp], g_src [V,h,w,Cp])."""
    src_nhwc = _need(src_nhwc, "src_nhwc")
    V, h, w, Cp = src_nhwc.shape
    ref_nhwc = _need(ref_nhwc, "ref_nhwc", (h, w, Cp))
    d_candi = _need(d_candi, "d_candi")
    D = d_candi.numel()
    g_cost = _need(g_cost, "g_cost", (D, h, w))
    g_ref, g_src = torch.empty_like(ref_nhwc), torch.empty_like(src_nhwc)
    with torch.cuda.device(src_nhwc.device):
        nbytes = ctypes.c_size_t(0)
        _lib.check(_lib.load().nrgbd_costvol_bwd_workspace(V, Cp, D, h, w, ctypes.byref(nbytes)), "nrgbd_costvol_bwd_workspace")
        work = torch.empty(nbytes.value, dtype=torch.uint8, device=src_nhwc.device) if nbytes.value else None
        rc = _lib.load().nrgbd_costvol_bwd(_p(ref_nhwc), _p(src_nhwc), _p(_need(KR, "KR").reshape(V, 9)), _p(_need(Kt, "Kt", (V, 3))),
                                           _p(_need(rays, "rays", (3, h * w))), _p(d_candi), float(cx), float(cy), float(sigma),
                                           DIST[dist], int(bool(align_corners)), _p(g_cost), _p(g_ref), _p(g_src),
                                           V, int(C), Cp, D, h, w, _p(work) if work is not None else None, nbytes.value,
                                           _stream(src_nhwc))
    _lib.check(rc, "nrgbd_costvol_bwd")
    return g_ref, g_src


def warp_volume(src, src_strides, ref, ref_strides, KR, Kt, rays, d_candi, cx, cy, V, Cs, h, w,
                bv_cur=None, bv_pred=None, align_corners=False, channels_last=False):
    """Plane-sweep warp with samples kept (+ K-Net input assembly) -> [V*Cs (+Cs) (+1), D, h, w]
    (or [D, h, w, channels] with channels_last=True, the layout conv3d consumes).

    `src` / `ref` are any CUDA fp32 tensors; `src_strides` = (view, channel, y, x) and
    `ref_strides` = (channel, y, x) element strides from their data pointers.
    """
    src = _need(src, "src", strided=True)  # addressed through src_strides
    KR = _need(KR, "KR").reshape(V, 9)
    Kt = _need(Kt, "Kt", (V, 3))
    rays = _need(rays, "rays", (3, h * w))
    d_candi = _need(d_candi, "d_candi")
    D = d_candi.numel()
    n_ch = V * Cs
    if ref is not None:
        ref = _need(ref, "ref", strided=True)
        n_ch += Cs
    else:
        ref_strides = (0, 0, 0)
    if bv_cur is not None:
        bv_cur = _need(bv_cur, "bv_cur").reshape(D, h, w)
        bv_pred = _need(bv_pred, "bv_pred").reshape(D, h, w)
        n_ch += 1
    shape = (D, h, w, n_ch) if channels_last else (n_ch, D, h, w)
    out = torch.empty(shape, dtype=torch.float32, device=src.device)
    with torch.cuda.device(src.device):
        rc = _lib.load().nrgbd_warp_volume(_p(src), *[int(s) for s in src_strides], _p(ref),
                                           *[int(s) for s in ref_strides], _p(KR), _p(Kt), _p(rays),
                                           _p(d_candi), float(cx), float(cy), int(bool(align_corners)),
                                           _p(bv_cur), _p(bv_pred), _p(out), V, Cs, D, h, w,
                                           int(bool(channels_last)), _stream(src))
    _lib.check(rc, "nrgbd_warp_volume")
    return out


def dpv_resample(dpv, T, rays, d_candi, tan_hh, tan_hv, z_half, z_radius, pad_value, clamp=(-1000.0, 0.0),
                 new_candi=False):
    """PREDICT: dpv [D,h,w], T [4,4] (device) -> resampled [len(d_candi),h,w].  `d_candi` = the depths of the OUTPUT planes
    (= the source's unless new_candi: resample_vol_cuda's d_candi_new form, nrgbd_dpv_resample_to)."""
    dpv = _need(dpv, "dpv")
    D, h, w = dpv.shape
    T = _need(T, "T").reshape(16)
    rays = _need(rays, "rays", (3, h * w))
    d_candi = _need(d_candi, "d_candi") if new_candi else _need(d_candi, "d_candi", (D,))
    if d_candi.dim() != 1:
        raise ValueError("d_candi must be 1-D")
    Do = d_candi.shape[0]
    out = torch.empty((Do, h, w), dtype=torch.float32, device=dpv.device)
    lo, hi = clamp if clamp is not None else (0.0, 0.0)
    with torch.cuda.device(dpv.device):
        rc = _lib.load().nrgbd_dpv_resample_to(_p(dpv), _p(T), _p(rays), _p(d_candi), float(tan_hh),
                                               float(tan_hv), float(z_half), float(z_radius),
                                               float(pad_value), int(clamp is not None), float(lo),
                                               float(hi), _p(out), D, Do, h, w, _stream(dpv))
    _lib.check(rc, "nrgbd_dpv_resample_to")
    return out


def logsoftmax_d(a, b=None, scale=1.0):
    """log_softmax over dim 0 of scale*a (+ b); a, b [D, ...]."""
    a = _need(a, "a")
    D = a.shape[0]
    n = a.numel() // D
    if b is not None:
        b = _need(b, "b", a.shape)
    out = torch.empty_like(a)
    with torch.cuda.device(a.device):
        rc = _lib.load().nrgbd_logsoftmax_d(_p(a), _p(b), float(scale), _p(out), D, n, _stream(a))
    _lib.check(rc, "nrgbd_logsoftmax_d")
    return out


def depth_regress(logp, d_candi, want_conf=True):
    """logp [D, ...] -> (expected depth [...], max log-prob [...])."""
    logp = _need(logp, "logp")
    D = logp.shape[0]
    n = logp.numel() // D
    d_candi = _need(d_candi, "d_candi", (D,))
    depth = torch.empty(logp.shape[1:], dtype=torch.float32, device=logp.device)
    conf = torch.empty_like(depth) if want_conf else None
    with torch.cuda.device(logp.device):
        rc = _lib.load().nrgbd_depth_regress(_p(logp), _p(d_candi), _p(depth), _p(conf), D, n, _stream(logp))
    _lib.check(rc, "nrgbd_depth_regress")
    return depth, conf


def export_depth_u16(logp, d_candi, depth_scale=1000.0, conf_scale=1000.0):
    """logp [D, ...] -> (depth f32, conf f32 = exp(max logp), depth_u16, conf_u16 [as int16 storage viewed uint16])."""
    logp = _need(logp, "logp")
    D = logp.shape[0]
    n = logp.numel() // D
    d_candi = _need(d_candi, "d_candi", (D,))
    shp = logp.shape[1:]
    depth = torch.empty(shp, dtype=torch.float32, device=logp.device)
    conf = torch.empty_like(depth)
    du = torch.empty(shp, dtype=torch.uint16, device=logp.device)
    cu = torch.empty(shp, dtype=torch.uint16, device=logp.device)
    with torch.cuda.device(logp.device):
        rc = _lib.load().nrgbd_export_depth_u16(_p(logp), _p(d_candi), float(depth_scale), float(conf_scale), _p(depth),
                                                 _p(conf), _p(du), _p(cu), D, n, _stream(logp))
    _lib.check(rc, "nrgbd_export_depth_u16")
    return depth, conf, du, cu


def warp_depth_fwd(src, dmap, K, R, t, rays):
    """src [N,C,H,W], dmap [H,W], K [3,3], R [N,3,3], t [N,3], rays [3,HW] -> warped [N,C,H,W] (nrgbd_warp_depth_fwd)."""
    src = _need(src, "src")
    N, C, H, W = src.shape
    dmap = _need(dmap, "dmap").reshape(H, W)
    K = _need(K, "K", (3, 3)); R = _need(R, "R", (N, 3, 3)); t = _need(t, "t", (N, 3)); rays = _need(rays, "rays", (3, H * W))
    out = torch.empty_like(src)
    with torch.cuda.device(src.device):
        rc = _lib.load().nrgbd_warp_depth_fwd(_p(src), _p(dmap), _p(K), _p(R), _p(t), _p(rays), _p(out), N, C, H, W, _stream(src))
    _lib.check(rc, "nrgbd_warp_depth_fwd")
    return out


def warp_depth_bwd(src, dmap, K, R, t, rays, g_out):
    """Gradient of sum(warp_depth_fwd(...) * g_out) w.r.t. (R [N,3,3], t [N,3])."""
    src = _need(src, "src")
    N, C, H, W = src.shape
    dmap = _need(dmap, "dmap").reshape(H, W)
    K = _need(K, "K", (3, 3)); R = _need(R, "R", (N, 3, 3)); t = _need(t, "t", (N, 3)); rays = _need(rays, "rays", (3, H * W))
    g_out = _need(g_out, "g_out", src.shape)
    nwg = int(_lib.load().nrgbd_warp_depth_bwd_workgroups(H, W))
    partial = torch.empty((N, nwg, 12), dtype=torch.float32, device=src.device)
    g_R = torch.empty((N, 3, 3), dtype=torch.float32, device=src.device)
    g_t = torch.empty((N, 3), dtype=torch.float32, device=src.device)
    with torch.cuda.device(src.device):
        rc = _lib.load().nrgbd_warp_depth_bwd(_p(src), _p(dmap), _p(K), _p(R), _p(t), _p(rays), _p(g_out), _p(partial),
                                               _p(g_R), _p(g_t), N, C, H, W, _stream(src))
    _lib.check(rc, "nrgbd_warp_depth_bwd")
    return g_R, g_t


# ----------------------------------------------------------------------------- K-Net convolutions
def conv3d_workgroups(D, H, W):
    return int(_lib.load().nrgbd_conv3d_workgroups(D, H, W))


def conv3d_pack_weights(w):
    """w [64, Cin, 3, 3, 3] -> packed B-operand stream for conv3d (one 1 KB line per wave load)."""
    w = _need(w, "w")
    cout, cin = w.shape[:2]
    if cout != 64 or tuple(w.shape[2:]) != (3, 3, 3):
        raise ValueError("conv3d_pack_weights expects [64, Cin, 3, 3, 3], got %s" % (tuple(w.shape),))
    wp = torch.empty(27 * cin * 64, dtype=torch.float32, device=w.device)
    with torch.cuda.device(w.device):
        rc = _lib.load().nrgbd_conv3d_pack_weights(_p(w), _p(wp), cin, _stream(w))
    _lib.check(rc, "nrgbd_conv3d_pack_weights")
    return wp


def conv3d(x, w_packed, x_ss=None, x_relu=False, res=None, res_ss=None, res_relu=False,
           materialize=False, want_stats=True, out=None, mat_out=None):
    """Channels-last 3x3x3 convolution on the fp32 matrix cores.

    x [D,H,W,Cin]; input = act(x*s+t) (+ act(res*s'+t')).  Returns (y [D,H,W,64], stats | None, materialized | None).
    """
    x = _need(x, "x")
    D, H, W, Cin = x.shape
    y = out if out is not None else torch.empty((D, H, W, 64), dtype=torch.float32, device=x.device)
    nwg = conv3d_workgroups(D, H, W)
    stats = torch.empty((nwg, 128), dtype=torch.float32, device=x.device) if want_stats else None
    mat = None
    if materialize:
        mat = mat_out if mat_out is not None else torch.empty_like(x)
    if res is not None:
        res = _need(res, "res", x.shape)
    with torch.cuda.device(x.device):
        rc = _lib.load().nrgbd_conv3d_3x3x3_f32(_p(x), _p(x_ss), int(x_relu), _p(res), _p(res_ss), int(res_relu),
                                                 _p(mat), _p(w_packed), _p(y), _p(stats), D, H, W, Cin, 64, _stream(x))
    _lib.check(rc, "nrgbd_conv3d_3x3x3_f32")
    return y, stats, mat


def conv_wino_pack(w, transposed=False):
    """w [Cout, Cin, 3, 3, 3] (kd = 3) or [Cout, Cin, 3, 3] (kd = 1) -> Winograd-domain B-operand stream of nrgbd_conv_wino_f32:
    U = G g G^T over (ky, kx) in float64, rounded once to fp32, laid out [cg][stage = cb*kd + depth tap][xi = 4*xi_y + xi_x]
    [wave][lane = kq*16 + j][e] with ci = cb*16 + 4*kq + e and co = cg*64 + 16*wave + j.
    transposed: w is the FORWARD weight [Cin', Cout', ...] of a layer and the stream is the one of its data gradient (channel
    axes swapped, taps flipped) — the kernel then maps dL/dy [.., Cin' of this call = the layer's Cout] to dL/dx."""
    w = _need(w, "w")
    if w.dim() == 4:
        w = w[:, :, None]
    if w.dim() != 5 or tuple(w.shape[3:]) != (3, 3) or w.shape[2] not in (1, 3):
        raise ValueError("conv_wino_pack expects [Cout, Cin, (3,) 3, 3], got %s" % (tuple(w.shape),))
    Cout, Cin, KD = (w.shape[1], w.shape[0], w.shape[2]) if transposed is True or transposed == 1 else w.shape[:3]
    if Cout % 64 or Cin % 16 or (transposed == 2 and Cin % 64):
        raise ValueError("conv_wino_pack: Cout %% 64 and Cin %% 16 required, got Cout=%d Cin=%d" % (Cout, Cin))
    # transposed = 2: both streams in one launch -> a buffer of twice the size, forward half first (training: ops.conv_wino_pack_both)
    wp = torch.empty(Cout * Cin * KD * 16 * (2 if transposed == 2 else 1), dtype=torch.float32, device=w.device)
    wc = w.detach().contiguous()
    with torch.cuda.device(w.device):
        rc = _lib.load().nrgbd_conv_wino_pack(_p(wc), _p(wp), Cin, Cout, KD, int(transposed), _stream(w))
    _lib.check(rc, "nrgbd_conv_wino_pack")
    return wp


def conv_wino_pack_both(w, dw=False):
    """(forward stream, data-gradient stream) of a layer's weight in ONE launch (training re-packs both every iteration).
    dw: the streams of wino_dw.hip (Winograd along depth too; dw = 4: wino_dw4.hip, F(4,3) along depth) instead of wino_pc.hip's."""
    both = (conv_wino_dw4_pack if dw == 4 else conv_wino_dw_pack if dw else conv_wino_pack)(w, transposed=2)
    n = both.numel() // 2
    return both[:n], both[n:]


def conv_wino_pack_reference(w):
    """The same stream through torch (einsum in float64): what the device packer is tested against."""
    if w.dim() == 4:
        w = w[:, :, None]
    Cout, Cin, KD = w.shape[:3]
    G = torch.tensor([[1.0, 0.0, 0.0], [0.5, 0.5, 0.5], [0.5, -0.5, 0.5], [0.0, 0.0, 1.0]], dtype=torch.float64, device=w.device)
    U = torch.einsum("ay,ockyx,bx->ockab", G, w.detach().double(), G).reshape(Cout, Cin, KD, 16)   # [co, ci, kd, xi]
    U = U.reshape(Cout // 64, 4, 16, Cin // 16, 4, 4, KD, 16)   # co -> (cg, wave, j); ci -> (cb, kq, e)
    U = U.permute(0, 3, 6, 7, 1, 4, 2, 5).contiguous()          # [cg, cb, kd, xi, wave, kq, j, e]
    return U.to(torch.float32).reshape(-1)


def conv_wino_supported(N, H, W, Cin, Cout, kd):
    """Shapes nrgbd_conv_wino_f32 accepts (the callers fall back to the direct kernels otherwise): channel multiples, at least
    two stages, an even stage count except for the 16-channel 3-D first layer, 32-bit byte offsets in the loader."""
    stages = (Cin // 16) * kd
    span = (1 if kd == 3 else N) * H * W * Cin
    if Cout == 32:      # the HALF form of wino_pc.hip: 2-D, dilation 1 (the caller's business), weights packed by conv_wino_pack32
        return kd == 1 and Cin % 16 == 0 and stages >= 2 and stages % 2 == 0 and span < (1 << 30)
    return Cin % 16 == 0 and Cout % 64 == 0 and stages >= 2 and (stages % 2 == 0 or (kd == 3 and Cin == 16)) and span < (1 << 30)


def conv_wino_pack32(w):
    """Weight stream of a 32-output-channel 3x3 layer for the HALF form of nrgbd_conv_wino_f32 (Cout = 32): the 64-column stream
    with the upper 32 columns zero (the kernel's waves read the two 16-column lines that exist)."""
    w = _need(w, "w")
    if w.dim() != 4 or w.shape[0] != 32 or tuple(w.shape[2:]) != (3, 3):
        raise ValueError("conv_wino_pack32 expects [32, Cin, 3, 3], got %s" % (tuple(w.shape),))
    return conv_wino_pack(torch.cat((w.detach(), torch.zeros_like(w)), 0))


def conv_wino_tiles(N, H, W, dilation=1):
    return int(_lib.load().nrgbd_conv_wino_tiles(N, H, W, dilation))


def conv_wino(x, w_wino, Cout, kd, dilation=1, x_ss=None, x_relu=False, res=None, res_ss=None, res_relu=False,
              materialize=False, want_stats=True):
    """Channels-last 3x3(x3) stride-1 convolution in the Winograd domain on the producer/consumer kernel (wino_pc.hip):
    x [N,H,W,Cin] -> (y [N,H,W,Cout], stats [2*Cout, tiles] (column-major partials for bn_finalize_cm) | None,
    materialized | None); kd = 3 convolves over N as depth."""
    x = _need(x, "x")
    N, H, W, Cin = x.shape
    y = torch.empty((N, H, W, Cout), dtype=torch.float32, device=x.device)
    half = Cout == 32     # HALF form: one statistics row per (tile, row block); the weight stream is the zero-padded 64-column one
    if half and (kd != 1 or dilation != 1):
        raise ValueError("conv_wino: Cout = 32 is the 2-D, dilation-1 form only")
    stats = torch.empty((2 * Cout, (2 if half else 1) * conv_wino_tiles(N, H, W, dilation)), dtype=torch.float32, device=x.device) if want_stats else None
    mat = torch.empty_like(x) if materialize else None
    if res is not None:
        res = _need(res, "res", x.shape)
    if w_wino.numel() != max(1, Cout // 64) * (Cin // 16) * kd * 16 * 1024:
        raise ValueError("conv_wino: packed weights do not match Cin=%d Cout=%d kd=%d" % (Cin, Cout, kd))
    with torch.cuda.device(x.device):
        rc = _lib.load().nrgbd_conv_wino_f32(_p(x), _p(x_ss), int(x_relu), _p(res), _p(res_ss), int(res_relu), _p(mat),
                                              _p(w_wino), _p(y), _p(stats), N, H, W, Cin, Cout, kd, dilation, _stream(x))
    _lib.check(rc, "nrgbd_conv_wino_f32")
    return y, stats, mat


def conv_wino_dw_pack(w, transposed=False):
    """w [Cout, Cin, 3, 3, 3] -> weight stream of nrgbd_conv_wino_dw_f32 (Winograd in depth too): U_t = sum_kd G[t][kd] (G g_kd G^T)
    in float64, rounded once, laid out [cg][stage = t*(Cin/16) + cb][xi][wave][lane = kq*16 + j][e]."""
    w = _need(w, "w")
    if w.dim() != 5 or tuple(w.shape[2:]) != (3, 3, 3):
        raise ValueError("conv_wino_dw_pack expects [Cout, Cin, 3, 3, 3], got %s" % (tuple(w.shape),))
    Cout, Cin = (w.shape[1], w.shape[0]) if transposed is True or transposed == 1 else w.shape[:2]
    if Cout % 64 or Cin % 16 or (transposed == 2 and Cin % 64):
        raise ValueError("conv_wino_dw_pack: Cout %% 64 and Cin %% 16 required, got Cout=%d Cin=%d" % (Cout, Cin))
    wp = torch.empty(Cout * Cin * 4 * 16 * (2 if transposed == 2 else 1), dtype=torch.float32, device=w.device)
    wc = w.detach().contiguous()
    with torch.cuda.device(w.device):
        rc = _lib.load().nrgbd_conv_wino_dw_pack(_p(wc), _p(wp), Cin, Cout, int(transposed), _stream(w))
    _lib.check(rc, "nrgbd_conv_wino_dw_pack")
    return wp


def conv_wino_dw_pack_reference(w):
    """The same stream through torch (einsum in float64): what the device packer is tested against."""
    Cout, Cin = w.shape[:2]
    G = torch.tensor([[1.0, 0.0, 0.0], [0.5, 0.5, 0.5], [0.5, -0.5, 0.5], [0.0, 0.0, 1.0]], dtype=torch.float64, device=w.device)
    U = torch.einsum("tz,ay,oczyx,bx->octab", G, G, w.detach().double(), G).reshape(Cout, Cin, 4, 16)   # [co, ci, t, xi]
    U = U.reshape(Cout // 64, 4, 16, Cin // 16, 4, 4, 4, 16)    # co -> (cg, wave, j); ci -> (cb, kq, e); t; xi
    U = U.permute(0, 6, 3, 7, 1, 4, 2, 5).contiguous()          # [cg, t, cb, xi, wave, kq, j, e]
    return U.to(torch.float32).reshape(-1)


def conv_wino_dw_supported(N, H, W, Cin, Cout):
    """Shapes nrgbd_conv_wino_dw_f32 accepts: pairs of depth slices, whole 8x16 tiles, 32-bit in-plane byte offsets."""
    return N % 2 == 0 and N >= 2 and H % 8 == 0 and W % 16 == 0 and Cin % 16 == 0 and Cout % 64 == 0 and H * W * Cin < (1 << 30)


def relu_unit(gamma, beta, n):
    """2^-k with 2^k > |gamma_c| sqrt(n) + |beta_c| for every channel: a bound on |BatchNorm(y)_c| under batch statistics over n
    values that holds for ANY data (|y - mean| <= sqrt(n var)), i.e. the x_unit of the clamped-FMA convolution forms.  Reads the
    affine parameters once (host synchronisation: callers cache it with the packed weights)."""
    import math
    bound = float(gamma.detach().abs().max()) * math.sqrt(float(n)) + float(beta.detach().abs().max())
    # One extra power of two of headroom (free: a power-of-two unit changes no bit below saturation).  The bound is exact for exact
    # statistics; the kernels finalise the variance as E[y^2] - mean^2 from fp32 per-tile sums (in float64), which can UNDER-estimate
    # it when |mean| >> std (a tile's partial of y^2 is good to ~1.5e-5 relative) and so over-scale the normalised values.  The
    # finalisers guard the other side (csrc/common.hpp bn_finalize_channel): a channel whose computed variance falls below
    # 1e-5 mean^2 gets a NaN scale (loud: the frame's outputs are NaN, valid_dpv trips) — whenever the scale is finite the computed
    # variance is at least 1 / 2.5 of the true one, inside the factor 4 that the doubled bound covers.  The clamp can therefore not
    # saturate silently (tests/test_gpu_knet.py constructs the collapsing channel)
    return 2.0 ** -max(0, math.floor(math.log2(max(2.0 * bound, 1e-30))) + 1)


def conv_wino_dw(x, w_wino, Cout, x_ss=None, x_relu=False, res=None, res_ss=None, res_relu=False, materialize=False,
                 want_stats=True, x_unit=0.0):
    """Channels-last 3x3x3 stride-1 convolution over x [D,H,W,Cin] with Winograd in all three dimensions (wino_dw.hip):
    -> (y [D,H,W,Cout], stats [2*Cout, tiles] (column-major partials for bn_finalize_cm) | None, materialized | None).
    x_unit = 2^-k > 0: the clamped-FMA form (nrgbd_conv_wino_dw_unit_f32) — x_ss and x_relu required, no residual / materialise,
    and w_wino packed from 2^k * w."""
    x = _need(x, "x")
    N, H, W, Cin = x.shape
    y = torch.empty((N, H, W, Cout), dtype=torch.float32, device=x.device)
    stats = torch.empty((2 * Cout, conv_wino_tiles(N, H, W, 1)), dtype=torch.float32, device=x.device) if want_stats else None
    mat = torch.empty_like(x) if materialize else None
    if res is not None:
        res = _need(res, "res", x.shape)
    if w_wino.numel() != (Cout // 64) * (Cin // 16) * 4 * 16 * 1024:
        raise ValueError("conv_wino_dw: packed weights do not match Cin=%d Cout=%d" % (Cin, Cout))
    with torch.cuda.device(x.device):
        if x_unit:
            if x_ss is None or not x_relu or res is not None or materialize:
                raise ValueError("conv_wino_dw: x_unit is the plain BatchNorm + ReLU form only")
            rc = _lib.load().nrgbd_conv_wino_dw_unit_f32(_p(x), _p(x_ss), float(x_unit), _p(w_wino), _p(y), _p(stats), N, H, W, Cin,
                                                          Cout, _stream(x))
        else:
            rc = _lib.load().nrgbd_conv_wino_dw_f32(_p(x), _p(x_ss), int(x_relu), _p(res), _p(res_ss), int(res_relu), _p(mat),
                                                     _p(w_wino), _p(y), _p(stats), N, H, W, Cin, Cout, _stream(x))
    _lib.check(rc, "nrgbd_conv_wino_dw_f32")
    return y, stats, mat


def conv_wino_dw4_supported(N, H, W, Cin, Cout):
    """Shapes nrgbd_conv_wino_dw4_f32 accepts: quadruples of depth slices, whole 8x16 tiles, 32-bit in-plane byte offsets."""
    return N % 4 == 0 and N >= 4 and H % 8 == 0 and W % 16 == 0 and Cin % 16 == 0 and Cout % 64 == 0 and H * W * Cin < (1 << 30)


def conv_wino_dw4_pack(w, transposed=False):
    """w [Cout, Cin, 3, 3, 3] -> weight stream of nrgbd_conv_wino_dw4_f32 (F(2x2,3x3) in the plane x F(4,3) along depth, points
    0, +-1/2, +-3/2, inf): U_t = sum_kd Gd[t][kd] (G g_kd G^T) in float64, rounded once, phases in execution order t = 1,2,3,4,0,5.
    transposed: True = the data-gradient stream (transposed + flipped weights), 2 = both streams (forward, then data gradient)."""
    w = _need(w, "w")
    if w.dim() != 5 or tuple(w.shape[2:]) != (3, 3, 3):
        raise ValueError("conv_wino_dw4_pack expects [Cout, Cin, 3, 3, 3], got %s" % (tuple(w.shape),))
    Cout, Cin = (w.shape[1], w.shape[0]) if transposed is True or transposed == 1 else w.shape[:2]
    if Cout % 64 or Cin % 16 or (transposed == 2 and Cin % 64):
        raise ValueError("conv_wino_dw4_pack: Cout %% 64 and Cin %% 16 required, got Cout=%d Cin=%d" % (Cout, Cin))
    wp = torch.empty(Cout * Cin * 6 * 16 * (2 if transposed == 2 else 1), dtype=torch.float32, device=w.device)
    with torch.cuda.device(w.device):
        rc = _lib.load().nrgbd_conv_wino_dw4_pack(_p(w.detach().contiguous()), _p(wp), Cin, Cout, int(transposed), _stream(w))
    _lib.check(rc, "nrgbd_conv_wino_dw4_pack")
    return wp


def conv_wino_dw4(x, w_wino, Cout, x_ss=None, x_relu=False, want_stats=True, x_unit=0.0):
    """Channels-last 3x3x3 stride-1 convolution over x [D,H,W,Cin] with F(4,3) along depth on top of the in-plane F(2x2,3x3)
    (wino_dw4.hip; D % 4 == 0): -> (y [D,H,W,Cout], stats [2*Cout, tiles] | None).  Input forms: x as it is (x_ss None, x_relu False),
    act(x * s + t), or — x_unit = 2^-k > 0 — relu(x * s + t) as the clamped FMA with w_wino packed from 2^k * w."""
    x = _need(x, "x")
    N, H, W, Cin = x.shape
    if w_wino.numel() != (Cout // 64) * (Cin // 16) * 6 * 16 * 1024:
        raise ValueError("conv_wino_dw4: packed weights do not match Cin=%d Cout=%d" % (Cin, Cout))
    y = torch.empty((N, H, W, Cout), dtype=torch.float32, device=x.device)
    stats = torch.empty((2 * Cout, conv_wino_tiles(N, H, W, 1)), dtype=torch.float32, device=x.device) if want_stats else None
    lib = _lib.load()
    with torch.cuda.device(x.device):
        nbytes = ctypes.c_size_t(0)
        _lib.check(lib.nrgbd_conv_wino_dw4_workspace(N, H, W, Cout, ctypes.byref(nbytes)), "nrgbd_conv_wino_dw4_workspace")
        # per-launch scratch from the caching allocator (stream-aware: concurrent launches on other streams get other blocks; inside a
        # hipGraph capture it lives in the graph's pool): 32 KB per workgroup, no kernel is launched for it
        ws = torch.empty(nbytes.value, dtype=torch.uint8, device=x.device)
        rc = lib.nrgbd_conv_wino_dw4_f32(_p(x), _p(x_ss), int(x_relu), float(x_unit), _p(w_wino), _p(y), _p(stats), _p(ws),
                                         ctypes.c_size_t(ws.numel()), N, H, W, Cin, Cout, _stream(x))
    _lib.check(rc, "nrgbd_conv_wino_dw4_f32")
    return y, stats


def conv3d_wgrad(x, gy):
    """Weight gradient of the channels-last 3x3x3 convolution: x [D,H,W,Cin], gy [D,H,W,64] -> dW [64,Cin,3,3,3]."""
    x = _need(x, "x")
    D, H, W, Cin = x.shape
    gy = _need(gy, "gy", (D, H, W, 64))
    nwg = int(_lib.load().nrgbd_conv3d_wgrad_workgroups())
    partial = torch.empty(nwg * 27 * 64 * Cin, dtype=torch.float32, device=x.device)
    dw = torch.empty((64, Cin, 3, 3, 3), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        rc = _lib.load().nrgbd_conv3d_wgrad_f32(_p(x), _p(gy), _p(partial), _p(dw), D, H, W, Cin, _stream(x))
    _lib.check(rc, "nrgbd_conv3d_wgrad_f32")
    return dw


def conv3d_cout1(x, w_tap_major, x_ss=None, x_relu=False, res=None, res_ss=None, res_relu=False):
    """Last K-Net layer: x [D,H,W,64] -> y [D,H,W]; w_tap_major [27,64]."""
    x = _need(x, "x")
    D, H, W, Cin = x.shape
    w_tap_major = _need(w_tap_major, "w_tap_major", (27, Cin))
    y = torch.empty((D, H, W), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        rc = _lib.load().nrgbd_conv3d_3x3x3_cout1_f32(_p(x), _p(x_ss), int(x_relu), _p(res), _p(res_ss), int(res_relu),
                                                       _p(w_tap_major), _p(y), D, H, W, Cin, _stream(x))
    _lib.check(rc, "nrgbd_conv3d_3x3x3_cout1_f32")
    return y


def conv3d_cout1_dgrad(gy, w_tap_major):
    """Data gradient of the K-Net's last layer Conv3d(64, 1): gy [D,H,W], w_tap_major [27,64] -> gx [D,H,W,64] (conv3d_c1_bwd.hip)."""
    gy = _need(gy, "gy")
    D, H, W = gy.shape
    w_tap_major = _need(w_tap_major, "w_tap_major", (27, 64))
    gx = torch.empty((D, H, W, 64), dtype=torch.float32, device=gy.device)
    with torch.cuda.device(gy.device):
        rc = _lib.load().nrgbd_conv3d_cout1_dgrad_f32(_p(gy), _p(w_tap_major), _p(gx), D, H, W, _stream(gy))
    _lib.check(rc, "nrgbd_conv3d_cout1_dgrad_f32")
    return gx


def conv3d_cout1_wgrad(x, gy):
    """Weight gradient of the K-Net's last layer Conv3d(64, 1): x [D,H,W,64], gy [D,H,W] -> dW [1,64,3,3,3] (conv3d_c1_bwd.hip)."""
    x = _need(x, "x")
    D, H, W, Cin = x.shape
    if Cin != 64:
        raise ValueError("conv3d_cout1_wgrad: 64 input channels, got %d" % Cin)
    gy = _need(gy, "gy", (D, H, W))
    lib = _lib.load()
    dw = torch.empty((1, 64, 3, 3, 3), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        nbytes = ctypes.c_size_t(0)
        _lib.check(lib.nrgbd_conv3d_cout1_wgrad_workspace(D, H, W, ctypes.byref(nbytes)), "nrgbd_conv3d_cout1_wgrad_workspace")
        ws = torch.empty(nbytes.value, dtype=torch.uint8, device=x.device)
        rc = lib.nrgbd_conv3d_cout1_wgrad_f32(_p(x), _p(gy), _p(dw), _p(ws), ctypes.c_size_t(ws.numel()), D, H, W, _stream(x))
    _lib.check(rc, "nrgbd_conv3d_cout1_wgrad_f32")
    return dw


def _status_ptr(status, like):
    if status is None:
        return ctypes.c_void_p(0)
    if not (isinstance(status, torch.Tensor) and status.is_cuda and status.dtype == torch.int32 and status.numel() >= 1 and status.device == like.device):
        raise TypeError("status must be an int32 tensor on the device of the statistics")
    return ctypes.c_void_p(status.data_ptr())


def _nbt_ptr(nbt, like):
    if nbt is None:
        return ctypes.c_void_p(0)
    if not (isinstance(nbt, torch.Tensor) and nbt.is_cuda and nbt.dtype == torch.int64 and nbt.numel() == 1 and nbt.device == like.device):
        raise TypeError("batches_tracked must be an int64 scalar tensor on the device of the statistics")
    return ctypes.c_void_p(nbt.data_ptr())


def bn3d_finalize(stats, count, gamma, beta, eps, momentum, running_mean=None, running_var=None, status=None, batches_tracked=None):
    """Per-workgroup partials -> scale_shift [64,2]; updates the running statistics in place (train mode).
    status: int32 device word that counts variance-collapsed channels (include/nrgbd.h; nets.check_status raises on it)."""
    stats = _need(stats, "stats")
    ss = torch.empty((64, 2), dtype=torch.float32, device=stats.device)
    with torch.cuda.device(stats.device):
        rc = _lib.load().nrgbd_bn3d_finalize(_p(stats), stats.shape[0], int(count), _p(gamma), _p(beta), float(eps),
                                             float(momentum), _p(running_mean), _p(running_var), _p(ss), _status_ptr(status, stats), _nbt_ptr(batches_tracked, stats),
                                             _stream(stats))
    _lib.check(rc, "nrgbd_bn3d_finalize")
    return ss


# ----------------------------------------------------------------------------- 2-D feature CNN helpers
def conv2d_workgroups(N, H, W):
    return int(_lib.load().nrgbd_conv2d_workgroups(N, H, W))


def conv_pack_weights(w):
    """w [Cout, Cin, 3, 3] (or [Cout, Cin, 3, 3, 3]) -> packed B-operand stream of the matrix-core conv kernels."""
    w = _need(w, "w")
    cout, cin = w.shape[:2]
    taps = 1
    for k in w.shape[2:]:
        taps *= int(k)
    wp = torch.empty(taps * cin * cout, dtype=torch.float32, device=w.device)
    with torch.cuda.device(w.device):
        rc = _lib.load().nrgbd_conv_pack_weights(_p(w), _p(wp), cin, cout, taps, _stream(w))
    _lib.check(rc, "nrgbd_conv_pack_weights")
    return wp


def conv2d(x, w_packed, cout, dilation=1, x_ss=None, x_relu=False, res=None, res_ss=None, res_relu=False,
           materialize=False, bias=None, out_lrelu=False, want_stats=True):
    """Channels-last 3x3 convolution (stride 1, pad = dilation) on the fp32 matrix cores.

    x [N,H,W,Cin]; input = act(x*s+t) (+ act(res*s'+t')).  Returns (y [N,H,W,cout], stats | None, materialized | None).
    """
    x = _need(x, "x")
    N, H, W, Cin = x.shape
    y = torch.empty((N, H, W, cout), dtype=torch.float32, device=x.device)
    stats = torch.empty((conv2d_workgroups(N, H, W), 2 * cout), dtype=torch.float32, device=x.device) if want_stats else None
    mat = torch.empty_like(x) if materialize else None
    if res is not None:
        res = _need(res, "res", x.shape)
    with torch.cuda.device(x.device):
        rc = _lib.load().nrgbd_conv2d_3x3_f32(_p(x), _p(x_ss), int(x_relu), _p(res), _p(res_ss), int(res_relu), _p(mat),
                                               _p(w_packed), _p(bias), int(out_lrelu), _p(y), _p(stats), N, H, W, Cin,
                                               int(cout), int(dilation), _stream(x))
    _lib.check(rc, "nrgbd_conv2d_3x3_f32")
    return y, stats, mat


def conv_s2_pack(w, cp=None):
    """Weights of a stride-2, pad-1 3x3 convolution [Cout, C, 3, 3] -> packed stream of the equivalent 2x2-tap convolution on the
    space-to-depth input (nrgbd_conv2d_taps_f32, taps = 4): w2[co, (py*2+px)*C + c, ty, tx] with (ty, py) -> ky: (0,1) -> 0,
    (1,0) -> 1, (1,1) -> 2 and the same for x; (0,0) does not occur (zero).  cp = padded channel count (multiple of 16)."""
    w = _need(w, "w")
    return conv_pack_weights(conv_s2_weights(w, cp))


def conv_s2_weights(w, cp=None):
    """The re-indexing itself (any device): w [Cout, C, 3, 3] of a stride-2, pad-1 convolution -> w2 [Cout, cp, 2, 2] of the
    2x2-window convolution (window rows {y-1, y}, columns {x-1, x}) on the space-to-depth input."""
    cout, c = w.shape[:2]
    cp = cp or -(-4 * c // 16) * 16
    w2 = torch.zeros((cout, cp, 2, 2), dtype=w.dtype, device=w.device)
    kmap = {(0, 1): 0, (1, 0): 1, (1, 1): 2}
    for (ty, py), ky in kmap.items():
        for (tx, px), kx in kmap.items():
            ph = py * 2 + px
            w2[:, ph * c:(ph + 1) * c, ty, tx] = w[:, :, ky, kx]
    return w2


def space_to_depth2(x, nchw=False, cp=None):
    """[N,C,H,W] (nchw) or [N,H,W,C] -> [N,H/2,W/2,cp] with channel (py*2+px)*C + c = x[2y+py, 2x+px, c], zero-padded to cp."""
    x = _need(x, "x")
    if nchw:
        N, C, H, W = x.shape
    else:
        N, H, W, C = x.shape
    cp = cp or -(-4 * C // 16) * 16
    y = torch.empty((N, H // 2, W // 2, cp), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        rc = _lib.load().nrgbd_space_to_depth2(_p(x), int(nchw), _p(y), N, C, H, W, cp, _stream(x))
    _lib.check(rc, "nrgbd_space_to_depth2")
    return y


def conv2d_taps(x, w_packed, cout, taps, x_ss=None, x_relu=False, want_stats=True, stride=1):
    """1x1 convolution (taps = 1) or the 2x2-window form of a stride-2 3x3 convolution on a space-to-depth input (taps = 4) on
    the matrix-core kernel, with the trunk's prologue and statistics epilogue: x [N,H,W,Cin] -> (y [N,H,W,cout], stats | None).
    stride > 1 (taps = 1 only): the 1x1 convolution's own stride — x [N,H*s,W*s,Cin] is read at every s-th pixel."""
    x = _need(x, "x")
    N, H, W, Cin = x.shape
    if stride != 1:
        if taps != 1 or H % stride or W % stride:
            raise ValueError("conv2d_taps: a stride needs taps = 1 and sides that are multiples of it")
        H, W = H // stride, W // stride
    y = torch.empty((N, H, W, cout), dtype=torch.float32, device=x.device)
    stats = torch.empty((conv2d_workgroups(N, H, W), 2 * cout), dtype=torch.float32, device=x.device) if want_stats else None
    with torch.cuda.device(x.device):
        rc = _lib.load().nrgbd_conv2d_taps_f32(_p(x), _p(x_ss), int(x_relu), _p(w_packed), _p(y), _p(stats), N, H, W, Cin,
                                                int(cout), int(taps), int(stride), _stream(x))
    _lib.check(rc, "nrgbd_conv2d_taps_f32")
    return y, stats


def conv2d_wgrad(x, gy, dilation=1):
    """Weight gradient of the channels-last 3x3 stride-1 convolution: x [N,H,W,Cin], gy [N,H,W,Cout] -> dW [Cout,Cin,3,3]."""
    x = _need(x, "x")
    N, H, W, Cin = x.shape
    gy = _need(gy, "gy")
    Cout = gy.shape[-1]
    if tuple(gy.shape[:3]) != (N, H, W):
        raise ValueError("conv2d_wgrad: gy %s does not match x %s" % (tuple(gy.shape), tuple(x.shape)))
    lib = _lib.load()
    nwg = int(lib.nrgbd_conv2d_wgrad_workgroups(N, H, W, Cin, Cout))
    _lib.check(min(nwg, 0), "nrgbd_conv2d_wgrad_workgroups")
    blocks = -(-Cout // 64) * -(-Cin // 64)
    partial = torch.empty(blocks * nwg * 9 * 64 * 64, dtype=torch.float32, device=x.device)
    dw = torch.empty((Cout, Cin, 3, 3), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        rc = lib.nrgbd_conv2d_wgrad_f32(_p(x), _p(gy), _p(partial), _p(dw), N, H, W, Cin, Cout, int(dilation), _stream(x))
    _lib.check(rc, "nrgbd_conv2d_wgrad_f32")
    return dw


def conv2d_rnet(x, w_packed, cout, bias=None, lrelu=True, out=None, ldy=None, ycoff=0, cout_valid=None, mode=0, pa=0, pb=0):
    """R-Net layer on the matrix cores (nrgbd_conv2d_rnet_f32).  x [N,H,W,Cin] channels-last.
    mode 0: 3x3 conv -> out[..., ycoff:ycoff+cout_valid] of a [N,H,W,ldy] buffer (allocated [N,H,W,cout_valid] if None);
    mode 1: sub-pixel phase (pa, pb) of ConvTranspose2d(k4,s2,p1) -> the same inside a [N,2H,2W,ldy] buffer (required);
    mode 2: conv + bias + log_softmax over the channels -> planar [N,cout,H,W];
    mode 3: all four phases of the transposed conv in one launch (w_packed = the phases' packed weights concatenated)."""
    x = _need(x, "x")
    N, H, W, Cin = x.shape
    cv = cout if cout_valid is None else cout_valid
    if mode == 2:
        out = torch.empty((N, cout, H, W), dtype=torch.float32, device=x.device)
        ldy = cout
    elif out is None:
        if mode in (1, 3):
            raise ValueError("conv2d_rnet: a transposed conv writes into a caller-provided [N,2H,2W,ldy] buffer")
        out = torch.empty((N, H, W, cv), dtype=torch.float32, device=x.device)
        ldy = cv
    elif ldy is None:
        ldy = out.shape[-1]
    with torch.cuda.device(x.device):
        rc = _lib.load().nrgbd_conv2d_rnet_f32(_p(x), _p(w_packed), _p(bias), int(bool(lrelu)), _p(out), int(ldy), int(ycoff),
                                                int(cv), int(mode), int(pa), int(pb), N, H, W, Cin, int(cout), _stream(x))
    _lib.check(rc, "nrgbd_conv2d_rnet_f32")
    return out


def rnet_pack(dpv_log, feat, feat_planar, out=None):
    """dpv_log [D,h,w] (log-prob), feat [h,w,Cf] (or [Cf,h,w] if feat_planar) -> [1,h,w,D+Cf] = cat(exp(dpv), feat)."""
    dpv_log = _need(dpv_log, "dpv_log")
    feat = _need(feat, "feat")
    D, h, w = dpv_log.shape
    Cf = feat.shape[0] if feat_planar else feat.shape[-1]
    if out is None:
        out = torch.empty((1, h, w, D + Cf), dtype=torch.float32, device=dpv_log.device)
    elif tuple(out.shape) != (1, h, w, D + Cf) or not out.is_contiguous():
        raise ValueError("rnet_pack: out must be a contiguous [1,h,w,D+Cf] tensor")
    with torch.cuda.device(dpv_log.device):
        rc = _lib.load().nrgbd_rnet_pack(_p(dpv_log), _p(feat), int(bool(feat_planar)), _p(out), D, Cf, h * w, _stream(dpv_log))
    _lib.check(rc, "nrgbd_rnet_pack")
    return out


def bn_finalize(stats, count, gamma, beta, eps, momentum, running_mean=None, running_var=None, status=None, batches_tracked=None):
    """Per-workgroup partials [nwg, 2C] -> scale_shift [C,2]; updates the running statistics in place (train mode)."""
    stats = _need(stats, "stats")
    C = stats.shape[1] // 2
    ss = torch.empty((C, 2), dtype=torch.float32, device=stats.device)
    with torch.cuda.device(stats.device):
        rc = _lib.load().nrgbd_bn_finalize(_p(stats), stats.shape[0], C, int(count), _p(gamma), _p(beta), float(eps),
                                           float(momentum), _p(running_mean), _p(running_var), _p(ss), _status_ptr(status, stats), _nbt_ptr(batches_tracked, stats),
                                           _stream(stats))
    _lib.check(rc, "nrgbd_bn_finalize")
    return ss


def conv_wino_rnet(x, w_wino, cout, bias=None, lrelu=True, out=None, ycoff=0, cout_valid=None):
    """R-Net conv2d_leakyRelu block in the Winograd domain: x [N,H,W,Cin] -> leaky_relu(conv3x3(x) + bias).
    `cout` = columns of the packed weights (% 64, or 32: the HALF form, whose stream is the 64-column one with the upper half zero);
    `out` [N,H,W,ldy] may be wider than the layer (a concat buffer): output
    column c < cout_valid of a pixel lands at out[..., ycoff + c], the rest of the pixel is left alone."""
    x = _need(x, "x")
    N, H, W, Cin = x.shape
    valid = int(cout if cout_valid is None else cout_valid)
    if out is None:
        out = torch.empty((N, H, W, valid), dtype=torch.float32, device=x.device)
    elif tuple(out.shape[:3]) != (N, H, W) or not out.is_contiguous() or out.shape[3] < ycoff + valid:
        raise ValueError("conv_wino_rnet: out must be a contiguous [N,H,W,>= ycoff + cout_valid] tensor")
    with torch.cuda.device(x.device):
        rc = _lib.load().nrgbd_conv_wino_rnet_ex_f32(_p(x), _p(w_wino), _p(bias), int(bool(lrelu)), _p(out), N, H, W, Cin, int(cout),
                                                      int(out.shape[3]), int(ycoff), valid, _stream(x))
    _lib.check(rc, "nrgbd_conv_wino_rnet_ex_f32")
    return out


def conv2d_few(x, w_few, bias=None, lrelu=True, out=None, ycoff=0):
    """3x3 convolution with 1-4 output channels on the vector ALUs (nrgbd_conv2d_few_f32): x [N,H,W,ldx] channels-last, w_few
    [Cin/16, 9, Cout, 16] (= w [Cout, Cin, 3, 3] with ci -> (block, lane) and (ky, kx) -> tap; Cin % 16 == 0 <= ldx) ->
    out[..., ycoff:ycoff+Cout] of a contiguous [N,H,W,ldy] buffer (allocated [N,H,W,Cout] if None)."""
    x = _need(x, "x")
    w_few = _need(w_few, "w_few")
    N, H, W, ldx = x.shape
    nblk, taps, cout, lanes = w_few.shape
    if taps != 9 or lanes != 16 or not 1 <= cout <= 4 or nblk * 16 > ldx:
        raise ValueError("conv2d_few: w_few %s for an input of %d channels" % (tuple(w_few.shape), ldx))
    if out is None:
        out = torch.empty((N, H, W, cout), dtype=torch.float32, device=x.device)
    elif tuple(out.shape[:3]) != (N, H, W) or not out.is_contiguous() or out.shape[3] < ycoff + cout:
        raise ValueError("conv2d_few: out must be a contiguous [N,H,W,>= ycoff + Cout] tensor")
    with torch.cuda.device(x.device):
        rc = _lib.load().nrgbd_conv2d_few_f32(_p(x), int(ldx), _p(w_few), _p(bias), int(bool(lrelu)), _p(out), int(out.shape[3]), int(ycoff),
                                               N, H, W, nblk * 16, cout, _stream(x))
    _lib.check(rc, "nrgbd_conv2d_few_f32")
    return out


def bn_cl_supported(rows, C):
    """Shapes of the channels-last train-mode BatchNorm kernels (bn_train.hip): 16-byte channel quads that tile a 256-lane workgroup."""
    return rows > 0 and 4 <= C <= 1024 and C % 4 == 0 and 256 % (C // 4) == 0


def bn_cl_fwd(x, gamma, beta, eps, relu, residual=None, momentum=0.0, running_mean=None, running_var=None):
    """y = act(batchnorm(x)) + residual with batch statistics over the rows of x [rows, C] (contiguous, channels last).
    Returns (y, coef [4,C] = scale, shift, mean, invstd); updates the running statistics in place when given."""
    x = _need(x, "x")
    rows, C = x.shape
    if residual is not None:
        residual = _need(residual, "residual", (rows, C))
    y = torch.empty_like(x)
    coef = torch.empty((4, C), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        lib = _lib.load()
        G = lib.nrgbd_bn_cl_workgroups(rows, C)
        _lib.check(min(G, 0), "nrgbd_bn_cl_workgroups")
        partial = torch.empty((G, 2 * C), dtype=torch.float32, device=x.device)
        rc = lib.nrgbd_bn_cl_fwd(_p(x), _p(residual), _p(_need(gamma, "gamma", (C,))), _p(_need(beta, "beta", (C,))), float(eps),
                                 float(momentum), _p(running_mean), _p(running_var), int(bool(relu)), _p(y), _p(coef), _p(partial),
                                 rows, C, _stream(x))
    _lib.check(rc, "nrgbd_bn_cl_fwd")
    return y, coef


def bn_cl_bwd(x, gy, coef, relu):
    """Backward of bn_cl_fwd w.r.t. (x, gamma, beta): gx [rows, C], g_gamma [C], g_beta [C] (the residual's gradient is gy)."""
    x = _need(x, "x")
    rows, C = x.shape
    gy = _need(gy, "gy", (rows, C))
    coef = _need(coef, "coef", (4, C))
    gx = torch.empty_like(x)
    gg = torch.empty((2, C), dtype=torch.float32, device=x.device)
    coef2 = torch.empty((2, C), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        lib = _lib.load()
        G = lib.nrgbd_bn_cl_workgroups(rows, C)
        _lib.check(min(G, 0), "nrgbd_bn_cl_workgroups")
        partial = torch.empty((G, 2 * C), dtype=torch.float32, device=x.device)
        rc = lib.nrgbd_bn_cl_bwd(_p(x), _p(gy), _p(coef), int(bool(relu)), _p(gx), _p(gg[0]), _p(gg[1]), _p(coef2), _p(partial),
                                 rows, C, _stream(x))
    _lib.check(rc, "nrgbd_bn_cl_bwd")
    return gx, gg[0], gg[1]


def bn_finalize_cm(stats, count, gamma, beta, eps, momentum, running_mean=None, running_var=None, status=None, batches_tracked=None):
    """Column-major per-tile partials [2C, rows] (conv_wino) -> scale_shift [C,2]; updates the running statistics in place."""
    stats = _need(stats, "stats")
    C = stats.shape[0] // 2
    ss = torch.empty((C, 2), dtype=torch.float32, device=stats.device)
    with torch.cuda.device(stats.device):
        rc = _lib.load().nrgbd_bn_finalize_cm(_p(stats), stats.shape[1], C, int(count), _p(gamma), _p(beta), float(eps),
                                              float(momentum), _p(running_mean), _p(running_var), _p(ss), _status_ptr(status, stats), _nbt_ptr(batches_tracked, stats),
                                              _stream(stats))
    _lib.check(rc, "nrgbd_bn_finalize_cm")
    return ss


def logsoftmax_rows(x, inplace=True):
    """log_softmax over the last (channel) axis of a contiguous channels-last tensor [..., C], C in {64, 128} (nrgbd_logsoftmax_rows)."""
    x = _need(x, "x")
    C = x.shape[-1]
    y = x if inplace else torch.empty_like(x)
    with torch.cuda.device(x.device):
        rc = _lib.load().nrgbd_logsoftmax_rows(_p(x), _p(y), x.numel() // C, int(C), _stream(x))
    _lib.check(rc, "nrgbd_logsoftmax_rows")
    return y


def logsoftmax_d_bwd(logp, g, scale=1.0):
    """Backward of logsoftmax_d: scale * (g - exp(logp) * sum_k g) on planar [D, ...] volumes (nrgbd_logsoftmax_d_bwd)."""
    logp = _need(logp, "logp")
    g = _need(g, "g", logp.shape)
    D = logp.shape[0]
    gz = torch.empty_like(logp)
    with torch.cuda.device(logp.device):
        rc = _lib.load().nrgbd_logsoftmax_d_bwd(_p(logp), _p(g), float(scale), _p(gz), D, logp.numel() // D, _stream(logp))
    _lib.check(rc, "nrgbd_logsoftmax_d_bwd")
    return gz


def logsoftmax_rows_bwd(y, g):
    """Backward of logsoftmax_rows on contiguous channels-last tensors [..., C], C in {64, 128} (nrgbd_logsoftmax_rows_bwd)."""
    y = _need(y, "y")
    g = _need(g, "g", y.shape)
    C = y.shape[-1]
    gx = torch.empty_like(y)
    with torch.cuda.device(y.device):
        rc = _lib.load().nrgbd_logsoftmax_rows_bwd(_p(y), _p(g), _p(gx), y.numel() // C, int(C), _stream(y))
    _lib.check(rc, "nrgbd_logsoftmax_rows_bwd")
    return gx


def _nll_target(target, n):
    if not (isinstance(target, torch.Tensor) and target.is_cuda and target.dtype == torch.int64):
        raise TypeError("target must be an int64 tensor on the GPU")
    if target.numel() != n:
        raise ValueError("target has %d elements, the volume %d pixels" % (target.numel(), n))
    return target if target.is_contiguous() else target.contiguous()


def nll_fwd(logp, target, ignore_index, channels_last):
    """Mean NLL of a [D, n] (planar) or [n, D] (channels_last) log-probability volume -> stat [2] = (loss, counted pixels)
    (nrgbd_nll_fwd; a device tensor: no host synchronisation)."""
    logp = _need(logp, "logp")
    D = logp.shape[-1] if channels_last else logp.shape[0]
    n = logp.numel() // D
    target = _nll_target(target, n)
    lib = _lib.load()
    partial = torch.empty((lib.nrgbd_nll_workgroups(n), 2), dtype=torch.float32, device=logp.device)
    stat = torch.empty(2, dtype=torch.float32, device=logp.device)
    with torch.cuda.device(logp.device):
        rc = lib.nrgbd_nll_fwd(_p(logp), ctypes.c_void_p(target.data_ptr()), int(ignore_index), int(D), n, int(bool(channels_last)),
                               _p(partial), _p(stat), _stream(logp))
    _lib.check(rc, "nrgbd_nll_fwd")
    return stat


def nll_bwd(target, ignore_index, g_out, stat, shape, channels_last):
    """Gradient of nll_fwd w.r.t. logp, in logp's layout `shape` ([D, ...] or [..., D]) (nrgbd_nll_bwd)."""
    g_out = _need(g_out, "g_out").reshape(1)
    stat = _need(stat, "stat", (2,))
    D = shape[-1] if channels_last else shape[0]
    g = torch.empty(tuple(shape), dtype=torch.float32, device=stat.device)
    n = g.numel() // D
    target = _nll_target(target, n)
    with torch.cuda.device(stat.device):
        rc = _lib.load().nrgbd_nll_bwd(ctypes.c_void_p(target.data_ptr()), int(ignore_index), _p(g_out), _p(stat), _p(g), int(D), n,
                                       int(bool(channels_last)), _stream(stat))
    _lib.check(rc, "nrgbd_nll_bwd")
    return g


def bias_lrelu_cl_fwd(x, bias, slope):
    """y = leaky_relu(x + bias, slope) on channels-last rows x [rows, C] (nrgbd_bias_lrelu_cl_fwd)."""
    x = _need(x, "x")
    rows, C = x.shape
    y = torch.empty_like(x)
    with torch.cuda.device(x.device):
        rc = _lib.load().nrgbd_bias_lrelu_cl_fwd(_p(x), _p(_need(bias, "bias", (C,))), float(slope), _p(y), rows, C, _stream(x))
    _lib.check(rc, "nrgbd_bias_lrelu_cl_fwd")
    return y


def bias_lrelu_cl_bwd(y, gy, slope):
    """Backward of bias_lrelu_cl_fwd from its OUTPUT y: (gx [rows, C], g_bias [C])."""
    y = _need(y, "y")
    rows, C = y.shape
    gy = _need(gy, "gy", (rows, C))
    gx = torch.empty_like(y)
    gb = torch.empty((C,), dtype=torch.float32, device=y.device)
    with torch.cuda.device(y.device):
        lib = _lib.load()
        G = lib.nrgbd_bias_lrelu_cl_workgroups(rows, C)
        _lib.check(min(G, 0), "nrgbd_bias_lrelu_cl_workgroups")
        partial = torch.empty((G, C), dtype=torch.float32, device=y.device)
        rc = lib.nrgbd_bias_lrelu_cl_bwd(_p(y), _p(gy), float(slope), _p(gx), _p(gb), _p(partial), rows, C, _stream(y))
    _lib.check(rc, "nrgbd_bias_lrelu_cl_bwd")
    return gx, gb


def upsample_bilinear_ac(x, H, W, backward=False):
    """Bilinear up-sampling (align_corners=True) of a channels-last map x [N,bh,bw,C] -> [N,H,W,C] (nrgbd_upsample_bilinear_ac);
    backward=True: x is the gradient of the output [N,H2,W2,C] and (H, W) the INPUT size -> gradient of the input [N,H,W,C]."""
    x = _need(x, "x")
    N, a, b, C = x.shape
    y = torch.empty((N, H, W, C), dtype=torch.float32, device=x.device)
    bh, bw, Ho, Wo = (H, W, a, b) if backward else (a, b, H, W)
    with torch.cuda.device(x.device):
        rc = _lib.load().nrgbd_upsample_bilinear_ac(_p(x), _p(y), N, bh, bw, Ho, Wo, C, int(backward), _stream(x))
    _lib.check(rc, "nrgbd_upsample_bilinear_ac")
    return y


def spp_concat(quarter, deep, branches):
    """[quarter | deep | up(relu(bn(z_0))) | ... | up(relu(bn(z_3)))] -> [N,h,w,Cq+Cd+4*Cb] in one pass (nrgbd_spp_concat).
    quarter [N,h,w,Cq], deep [N,h,w,Cd] channels-last; branches: four (z [N,bh,bw,Cb] raw 1x1-conv output, ss [Cb,2]) in the
    concat's order; up = bilinear, align_corners=True (psm_submodule.py:149-161)."""
    quarter, deep = _need(quarter, "quarter"), _need(deep, "deep")
    N, h, w, Cq = quarter.shape
    Cd = deep.shape[3]
    if tuple(deep.shape[:3]) != (N, h, w) or len(branches) != 4:
        raise ValueError("spp_concat: quarter %s / deep %s / %d branches" % (tuple(quarter.shape), tuple(deep.shape), len(branches)))
    Cb = branches[0][0].shape[3]
    args = []
    for z, ss in branches:
        z, ss = _need(z, "branch"), _need(ss, "branch scale/shift", (Cb, 2))
        if z.shape[0] != N or z.shape[3] != Cb:
            raise ValueError("spp_concat: branch %s" % (tuple(z.shape),))
        args += [_p(z), _p(ss), int(z.shape[1]), int(z.shape[2])]
    out = torch.empty((N, h, w, Cq + Cd + 4 * Cb), dtype=torch.float32, device=quarter.device)
    with torch.cuda.device(quarter.device):
        rc = _lib.load().nrgbd_spp_concat(_p(quarter), Cq, _p(deep), Cd, *args, Cb, _p(out), N, h, w, _stream(quarter))
    _lib.check(rc, "nrgbd_spp_concat")
    return out


def nhwc_stats(x):
    """Per-workgroup (sum, sum of squares) partials of a channels-last tensor [..., C] -> [nwg, 2C]."""
    x = _need(x, "x")
    C = x.shape[-1]
    P = x.numel() // C
    nwg = int(_lib.load().nrgbd_nhwc_stats_workgroups(P))
    stats = torch.empty((nwg, 2 * C), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        rc = _lib.load().nrgbd_nhwc_stats(_p(x), P, C, _p(stats), _stream(x))
    _lib.check(rc, "nrgbd_nhwc_stats")
    return stats


def nhwc_act(x, x_ss=None, x_relu=False, res=None, res_ss=None, res_relu=False, out=None, ldy=None):
    """y = act(x*s+t) (+ act(res*s'+t')) on channels-last [..., C]; `out` may be a wider buffer (pixel stride ldy)."""
    x = _need(x, "x")
    C = x.shape[-1]
    P = x.numel() // C
    if res is not None:
        res = _need(res, "res", x.shape)
    if out is None:
        out, ldy = torch.empty_like(x), C
    elif ldy is None:
        raise ValueError("nhwc_act: out needs its pixel stride ldy")
    with torch.cuda.device(x.device):
        rc = _lib.load().nrgbd_nhwc_act(_p(x), _p(x_ss), int(x_relu), _p(res), _p(res_ss), int(res_relu), _p(out), P, C,
                                         int(ldy), _stream(x))
    _lib.check(rc, "nrgbd_nhwc_act")
    return out


def bn2d_train_act(x, gamma, beta, eps, relu=False, residual=None, inplace=True, want_mean_var=False):
    """Train-mode BatchNorm2d + activation (+ residual) in two HBM passes.  x [N,C,H,W] -> (y, mean_var | None)."""
    x = _need(x, "x")
    N, C, H, W = x.shape
    if residual is not None:
        residual = _need(residual, "residual", x.shape)
    y = x if inplace else torch.empty_like(x)
    partial = torch.empty(int(_lib.load().nrgbd_bn2d_partial_floats(C)), dtype=torch.float32, device=x.device)
    mv = torch.empty((C, 2), dtype=torch.float32, device=x.device) if want_mean_var else None
    with torch.cuda.device(x.device):
        rc = _lib.load().nrgbd_bn2d_train_act(_p(x), _p(gamma), _p(beta), float(eps), int(bool(relu)), _p(residual),
                                               _p(y), _p(partial), _p(mv), N, C, H * W, _stream(x))
    _lib.check(rc, "nrgbd_bn2d_train_act")
    return y, mv


def avgpool8(x):
    """8x8 / stride-8 average pooling of [N,C,H,W]."""
    x = _need(x, "x")
    N, C, H, W = x.shape
    y = torch.empty((N, C, H // 8, W // 8), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        rc = _lib.load().nrgbd_avgpool8(_p(x), _p(y), N * C, H, W, _stream(x))
    _lib.check(rc, "nrgbd_avgpool8")
    return y


def avgpool_cl(x, k):
    """k x k / stride-k average pooling of a channels-last map x [N,H,W,C] -> [N,H//k,W//k,C] (nrgbd_avgpool_cl; floor like avg_pool2d)."""
    x = _need(x, "x")
    N, H, W, C = x.shape
    y = torch.empty((N, H // k, W // k, C), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        rc = _lib.load().nrgbd_avgpool_cl(_p(x), _p(y), N, H, W, C, int(k), _stream(x))
    _lib.check(rc, "nrgbd_avgpool_cl")
    return y


def scatter_channels(src, dst, coff):
    """dst[r, y, x, coff + c] = src[c, y, x] for every image r of the channels-last buffer dst [R,H,W,ldy]; src is any strided
    [C,H,W] view (NCHW planes or a permuted channels-last tensor) (nrgbd_scatter_channels)."""
    src = _need(src, "src", strided=True)
    dst = _need(dst, "dst", strided=True)
    C, H, W = src.shape
    if dst.dim() != 4 or tuple(dst.shape[1:3]) != (H, W) or not dst.is_contiguous() or dst.shape[3] < coff + C:
        raise ValueError("scatter_channels: dst must be a contiguous [R,%d,%d,>= %d] tensor" % (H, W, coff + C))
    with torch.cuda.device(src.device):
        rc = _lib.load().nrgbd_scatter_channels(_p(src), src.stride(0), src.stride(1), src.stride(2), C, H, W, _p(dst),
                                                 int(dst.shape[3]), int(coff), int(dst.shape[0]), int(dst.stride(0)), _stream(src))
    _lib.check(rc, "nrgbd_scatter_channels")
    return dst


def bias_act_(x, bias, slope):
    """In place x = leaky_relu(x + bias[c], slope) on a contiguous [N,C,H,W] tensor (slope 1 = bias add only)."""
    x = _need(x, "x", strided=True)
    if not x.is_contiguous():
        raise ValueError("bias_act_ needs a contiguous NCHW tensor")
    N, C, H, W = x.shape
    bias = _need(bias, "bias", (C,))
    with torch.cuda.device(x.device):
        rc = _lib.load().nrgbd_bias_act_nchw(_p(x), _p(bias), float(slope), N, C, H * W, _stream(x))
    _lib.check(rc, "nrgbd_bias_act_nchw")
    return x
