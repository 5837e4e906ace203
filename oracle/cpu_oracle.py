"""ctypes binding of oracle/nrgbd_oracle.c (numpy in, numpy out).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg — never by neuralrgbd_amd/ (the product path has no CPU fallback).
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SRC = os.path.join(_HERE, "nrgbd_oracle.c")
_OUT_DIR = os.path.join(_HERE, "_build")
_SO = os.path.join(_OUT_DIR, "libnrgbd_oracle.so")

_lib = None


def build(force=False):
    """gcc-compile the C restatement (one rounding per operation: -ffp-contract=off)."""
    os.makedirs(_OUT_DIR, exist_ok=True)
    if force or not os.path.isfile(_SO) or os.path.getmtime(_SO) < os.path.getmtime(_SRC):
        cmd = ["gcc", "-O2", "-ffp-contract=off", "-fopenmp", "-shared", "-fPIC",
               "-o", _SO, _SRC, "-lm"]
        subprocess.check_call(cmd)
    return _SO


def lib():
    global _lib
    if _lib is None:
        _lib = ctypes.CDLL(build())
    return _lib


def _f(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a, a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))


def set_threads(n):
    return int(lib().oracle_set_threads(int(n)))


def homography_terms(K, R, t):
    """(K R_v) [V,9] and K t_v [V,3] in the summation order of the reference's CPU matmuls (homography.py:315-317),
    written out in C (oracle_homography_terms) so that the result does not depend on the host's BLAS kernels."""
    k, pk = _f(np.asarray(K, np.float32).reshape(3, 3))
    r, pr = _f(np.asarray(R, np.float32).reshape(-1, 3, 3))
    tt, pt = _f(np.asarray(t, np.float32).reshape(-1, 3))
    V = r.shape[0]
    KR = np.empty((V, 9), np.float32)
    Kt = np.empty((V, 3), np.float32)
    rc = lib().oracle_homography_terms(pk, pr, pt, V, KR.ctypes.data_as(ctypes.POINTER(ctypes.c_float)),
                                       Kt.ctypes.data_as(ctypes.POINTER(ctypes.c_float)))
    assert rc == 0
    return KR, Kt


def pose_inverse(T):
    """T [...,4,4] -> inverse in the path's fixed fp64 Gauss-Jordan order, rounded to fp32 (oracle_pose_inverse; replaces the
    host-LAPACK `.inverse()` of test_utils/test_KVNet.py:50).  Raises on a singular matrix like the reference does."""
    t, pt = _f(np.asarray(T, np.float32))
    out = np.empty_like(t)
    bad = lib().oracle_pose_inverse(pt, out.ctypes.data_as(ctypes.POINTER(ctypes.c_float)), t.size // 16)
    if bad:
        raise np.linalg.LinAlgError("pose_inverse: singular matrix")
    return out


def costvol(feat_ref, feat_src, KR, Kt, rays, d_candi, cx, cy, sigma, dist="L2",
            align_corners=False):
    """feat_ref [C,h,w], feat_src [V,C,h,w] -> cost [D,h,w] (est_swp_volume_v4)."""
    V, C, h, w = feat_src.shape
    D = len(d_candi)
    fr, pfr = _f(feat_ref); fs, pfs = _f(feat_src)
    kr, pkr = _f(KR); kt, pkt = _f(Kt); ry, pry = _f(rays); dc, pdc = _f(d_candi)
    out = np.empty((D, h, w), np.float32)
    rc = lib().oracle_costvol(pfr, pfs, pkr, pkt, pry, pdc, ctypes.c_float(cx),
                              ctypes.c_float(cy), ctypes.c_float(sigma),
                              0 if dist == "L2" else 1, int(bool(align_corners)),
                              V, C, D, h, w, out.ctypes.data_as(ctypes.POINTER(ctypes.c_float)))
    assert rc == 0
    return out


def warp_volume(src, KR, Kt, rays, d_candi, cx, cy, align_corners=False):
    """src [V,Cs,h,w] -> [V,Cs,D,h,w] (warp_img_feats_v3)."""
    V, Cs, h, w = src.shape
    D = len(d_candi)
    s, ps = _f(src); kr, pkr = _f(KR); kt, pkt = _f(Kt); ry, pry = _f(rays); dc, pdc = _f(d_candi)
    out = np.empty((V, Cs, D, h, w), np.float32)
    rc = lib().oracle_warp_volume(ps, pkr, pkt, pry, pdc, ctypes.c_float(cx), ctypes.c_float(cy),
                                  int(bool(align_corners)), V, Cs, D, h, w,
                                  out.ctypes.data_as(ctypes.POINTER(ctypes.c_float)))
    assert rc == 0
    return out


def z_range(d_candi):
    """z_half, z_radius of homography.py:689-693 in fp32."""
    d32 = np.asarray(d_candi).astype(np.float32)
    z_max, z_min = d32.max(), d32.min()
    return np.float32((z_max + z_min) * np.float32(0.5)), np.float32((z_max - z_min) * np.float32(0.5))


def dpv_resample(dpv, T, rays, d_candi, tan_hh, tan_hv, pad, clamp=(-1000.0, 0.0), d_candi_new=None):
    """dpv [D,h,w], T [4,4] -> [D,h,w] (resample_vol_cuda + clamp); with d_candi_new -> [len(d_candi_new),h,w]
    (homography.py:675-693: points at the new candidates, z range = float64 min/max of d_candi cast to fp32)."""
    D, h, w = dpv.shape
    v, pv = _f(dpv); t, pt = _f(np.asarray(T).reshape(16)); ry, pry = _f(rays)
    if d_candi_new is None:
        dc, pdc = _f(d_candi)
        z_half, z_rad = z_range(d_candi)
    else:
        dc, pdc = _f(d_candi_new)
        d64 = np.asarray(d_candi)
        z_half, z_rad = float(np.float32((d64.max() + d64.min()) * .5)), float(np.float32((d64.max() - d64.min()) * .5))
    Do = dc.shape[0]
    out = np.empty((Do, h, w), np.float32)
    do_clamp = clamp is not None
    lo, hi = clamp if do_clamp else (0.0, 0.0)
    rc = lib().oracle_dpv_resample_to(pv, pt, pry, pdc, ctypes.c_float(tan_hh), ctypes.c_float(tan_hv),
                                      ctypes.c_float(z_half), ctypes.c_float(z_rad), ctypes.c_float(pad),
                                      int(do_clamp), ctypes.c_float(lo), ctypes.c_float(hi), D, Do, h, w,
                                      out.ctypes.data_as(ctypes.POINTER(ctypes.c_float)))
    assert rc == 0
    return out


def logsoftmax_d(a, b=None, scale=1.0):
    a_, pa = _f(a)
    D = a_.shape[0]
    n = a_.size // D
    pb = None
    if b is not None:
        b_, pb = _f(b)
    out = np.empty_like(a_)
    rc = lib().oracle_logsoftmax_d(pa, pb, ctypes.c_float(scale), D, ctypes.c_size_t(n),
                                   out.ctypes.data_as(ctypes.POINTER(ctypes.c_float)))
    assert rc == 0
    return out


def depth_regress(logp, d_candi):
    lp, plp = _f(logp); dc, pdc = _f(d_candi)
    D = lp.shape[0]
    n = lp.size // D
    depth = np.empty(lp.shape[1:], np.float32)
    conf = np.empty(lp.shape[1:], np.float32)
    rc = lib().oracle_depth_regress(plp, pdc, D, ctypes.c_size_t(n),
                                    depth.ctypes.data_as(ctypes.POINTER(ctypes.c_float)),
                                    conf.ctypes.data_as(ctypes.POINTER(ctypes.c_float)))
    assert rc == 0
    return depth, conf


def avgpool(img, pool):
    a, pa = _f(img)
    N, C, H, W = a.shape
    out = np.empty((N, C, H // pool, W // pool), np.float32)
    rc = lib().oracle_avgpool(pa, N, C, H, W, pool, out.ctypes.data_as(ctypes.POINTER(ctypes.c_float)))
    assert rc == 0
    return out


def export_depth_u16(logp, d_candi, depth_scale=1000.0, conf_scale=1000.0):
    """export_res_img (export_res.py:43-75): -> (depth f32, conf f32 = exp(max logp), depth_u16, conf_u16)."""
    lp, plp = _f(logp); dc, pdc = _f(d_candi)
    D = lp.shape[0]
    n = lp.size // D
    depth = np.empty(lp.shape[1:], np.float32); conf = np.empty(lp.shape[1:], np.float32)
    du = np.empty(lp.shape[1:], np.uint16); cu = np.empty(lp.shape[1:], np.uint16)
    rc = lib().oracle_export_depth_u16(plp, pdc, D, ctypes.c_size_t(n), ctypes.c_float(depth_scale), ctypes.c_float(conf_scale),
                                       depth.ctypes.data_as(ctypes.POINTER(ctypes.c_float)),
                                       conf.ctypes.data_as(ctypes.POINTER(ctypes.c_float)),
                                       du.ctypes.data_as(ctypes.POINTER(ctypes.c_ushort)),
                                       cu.ctypes.data_as(ctypes.POINTER(ctypes.c_ushort)))
    assert rc == 0
    return depth, conf, du, cu


def warp_depth_fwd(src, dmap, K, R, t, rays):
    """back_warp_th_Rt_msrc: src [N,C,H,W], dmap [H,W], K [3,3], R [N,3,3], t [N,3], rays [3,HW] -> [N,C,H,W]."""
    s_, ps = _f(src); d_, pd = _f(dmap); k_, pk = _f(K); r_, pr = _f(np.asarray(R, np.float32).reshape(-1, 9))
    t_, pt = _f(np.asarray(t, np.float32).reshape(-1, 3)); y_, py = _f(rays)
    N, C, H, W = s_.shape
    out = np.empty_like(s_)
    rc = lib().oracle_warp_depth_fwd(ps, pd, pk, pr, pt, py, N, C, H, W, out.ctypes.data_as(ctypes.POINTER(ctypes.c_float)))
    assert rc == 0
    return out


def warp_depth_bwd(src, dmap, K, R, t, rays, g_out):
    """d sum(out * g_out) / d(R, t) of warp_depth_fwd -> (g_R [N,3,3], g_t [N,3])."""
    s_, ps = _f(src); d_, pd = _f(dmap); k_, pk = _f(K); r_, pr = _f(np.asarray(R, np.float32).reshape(-1, 9))
    t_, pt = _f(np.asarray(t, np.float32).reshape(-1, 3)); y_, py = _f(rays); g_, pg = _f(g_out)
    N, C, H, W = s_.shape
    gR = np.empty((N, 3, 3), np.float32); gt = np.empty((N, 3), np.float32)
    rc = lib().oracle_warp_depth_bwd(ps, pd, pk, pr, pt, py, pg, N, C, H, W,
                                     gR.ctypes.data_as(ctypes.POINTER(ctypes.c_float)),
                                     gt.ctypes.data_as(ctypes.POINTER(ctypes.c_float)))
    assert rc == 0
    return gR, gt


def div_const_mismatches(c, stride=1):
    """Dividends for which the kernels' 3-instruction division by the constant c differs from IEEE a / c (must be 0)."""
    f = lib().oracle_div_const_mismatches
    f.restype = ctypes.c_long
    return int(f(ctypes.c_float(c), ctypes.c_long(stride)))
