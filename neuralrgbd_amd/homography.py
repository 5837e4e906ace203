"""Plane-sweep geometry operators — the reference's `warping.homography` call surface on HIP.

Same names, argument meaning and error behaviour as code/warping/homography.py, so that
`import neuralrgbd_amd.homography as warp_homo` drops into models/basic.py:271, models/KVNET.py:158-161
and test_utils/test_KVNet.py:54.  Every function launches kernels of libnrgbd_hip.so through
neuralrgbd_amd.ops; tensors must be on the GPU.

  est_swp_volume_v4      <- homography.py:293-331   (fused: no [D,C,h,w] warped tensor exists)
  warp_img_feats_v3      <- homography.py:234-280
  warp_img_feats_mgpu    <- homography.py:183-232
  resample_vol_cuda      <- homography.py:654-723   (analytic point grid, no host loop / H2D / sync)
  back_warp_th_Rt_msrc   <- homography.py:479-528   (LBA photometric warp through a depth map; differentiable in R, t)
  back_warp_th_Rt        <- homography.py:530-575   (its single-view form)
  get_rel_extrinsicM     <- homography.py:904-906
"""
import math

import numpy as np
import torch

from . import ops

# Device-resident copies of the per-trajectory constants.  The reference re-uploads the
# intrinsics, the ray table and d_candi on every call (homography.py:309-311, 247-249).
import collections

_const_cache = collections.OrderedDict()   # LRU: a long run builds a new intrinsics dict per trajectory / epoch
_CONST_CACHE_MAX = 64


def _cache_put(key, value):
    _const_cache[key] = value
    _const_cache.move_to_end(key)
    while len(_const_cache) > _CONST_CACHE_MAX:
        _const_cache.popitem(last=False)


def _d_candi_dev(d_candi, device):
    d32 = np.ascontiguousarray(np.asarray(d_candi).astype(np.float32))
    key = ("d", d32.tobytes(), str(device))
    hit = _const_cache.get(key)
    if hit is None:
        hit = torch.from_numpy(d32).to(device)
        _cache_put(key, hit)
    else:
        _const_cache.move_to_end(key)
    return hit


def _cam_dev(cam_intrinsic, device):
    """(K [3,3], rays [3,hw]) on `device`, cached per intrinsics dict (bounded LRU).  Dicts that carry only the
    reference's minimal keys ('unit_ray_array' [h,w,3] + 'intrinsic_M') get the 2-D / fp32 forms built here."""
    if "unit_ray_array_2D" not in cam_intrinsic and "unit_ray_array" in cam_intrinsic:
        ura = np.asarray(cam_intrinsic["unit_ray_array"])
        cam_intrinsic["unit_ray_array_2D"] = torch.from_numpy(ura.reshape(-1, 3).T.astype(np.float32).copy())
    if "intrinsic_M_cuda" not in cam_intrinsic and "intrinsic_M" in cam_intrinsic:
        cam_intrinsic["intrinsic_M_cuda"] = torch.from_numpy(np.asarray(cam_intrinsic["intrinsic_M"])[:3, :3].astype(np.float32))
    src = cam_intrinsic["unit_ray_array_2D"]
    key = ("cam", id(src), str(device))
    hit = _const_cache.get(key)
    if hit is None or hit[0] is not src:
        K = cam_intrinsic["intrinsic_M_cuda"].to(device=device, dtype=torch.float32).contiguous()
        rays = src.to(device=device, dtype=torch.float32).contiguous()
        hit = (src, K, rays)
        _cache_put(key, hit)
    else:
        _const_cache.move_to_end(key)
    return hit[1], hit[2]


def clear_cache():
    _const_cache.clear()


def cache_snapshot():
    """Strong references to every cached device constant (K, ray tables, d_candi).  A captured hipGraph bakes their raw
    pointers in and never touches the LRU on replay: the graph's owner keeps this list so that eviction (a long run builds a
    new intrinsics dict per trajectory) cannot hand the memory a live graph reads back to the allocator."""
    return list(_const_cache.values())


def homography_terms(K, R, t):
    """term1 = K t_v and the left factor K R_v of term2 (homography.py:315-317), batched: [V,3], [V,3,3]."""
    return ops.homography_terms(K, R, t)


def _stack(x):
    return torch.stack(list(x)) if isinstance(x, (list, tuple)) else x


def est_swp_volume_v4(feat_img_ref, feat_img_src, d_candi, R, t, cam_intrinsic, costV_sigma,
                      feat_dist="L2", debug_ipdb=False, align_corners=False):
    """Plane-sweep cost volume.

    feat_img_ref [1,C,h,w], feat_img_src [1,V,C,h,w], R [V,3,3], t [V,3] -> costV [1,D,h,w]
    with costV[d] = sum_v sum_c dist(warp_v,d(src)_c, ref_c) / costV_sigma.
    """
    if feat_dist not in ("L2", "L1"):
        raise Exception("undefined metric for feature distance ...")
    C, h, w = feat_img_ref.shape[1:]
    V = feat_img_src.shape[1]
    dev = feat_img_ref.device
    K, rays = _cam_dev(cam_intrinsic, dev)
    d_dev = _d_candi_dev(d_candi, dev)
    KR, Kt = homography_terms(K, _stack(R).reshape(V, 3, 3), _stack(t).reshape(V, 3))
    # NCHW -> 16-byte texels; inside the model the features are packed once by the D-Net instead
    texels = ops.pack_nhwc(torch.cat((feat_img_src[0], feat_img_ref), dim=0))
    cx, cy = cam_intrinsic["intrinsic_M"][0, 2], cam_intrinsic["intrinsic_M"][1, 2]
    cost, _ = ops.costvol(texels[V], texels[:V], KR, Kt, rays, d_dev, cx, cy, costV_sigma, C,
                          dist=feat_dist, align_corners=align_corners, want_cost=True, want_logp=False)
    return cost.unsqueeze(0)


def _warp_list(feat_img_src, d_candi, R, t, K, rays, cx, cy, align_corners):
    single = not (isinstance(R, list) and isinstance(t, list))
    maps = [feat_img_src] if single else list(feat_img_src)
    Rs = [R] if single else R
    ts = [t] if single else t
    V = len(maps)
    Cs, h, w = maps[0].shape[1:]
    dev = maps[0].device
    src = torch.cat(maps, dim=0).contiguous()  # [V,Cs,h,w]
    KR, Kt = homography_terms(K, torch.stack(list(Rs)).reshape(V, 3, 3), torch.stack(list(ts)).reshape(V, 3))
    vol = ops.warp_volume(src, (Cs * h * w, h * w, w, 1), None, None, KR, Kt, rays,
                          _d_candi_dev(d_candi, dev), cx, cy, V, Cs, h, w, align_corners=align_corners)
    views = [vol[v * Cs:(v + 1) * Cs] for v in range(V)]  # each [Cs,D,h,w]
    return views[0] if single else views


def warp_img_feats_v3(feat_img_src, d_candi, R, t, cam_intrinsic, align_corners=False):
    """Warp each source map to the reference view for all candidate depths.

    feat_img_src: list of [1,C,h,w] (or one tensor), R / t: lists (or single) -> list of [C,D,h,w].
    """
    first = feat_img_src[0] if isinstance(feat_img_src, (list, tuple)) else feat_img_src
    K, rays = _cam_dev(cam_intrinsic, first.device)
    cx, cy = cam_intrinsic["intrinsic_M"][0, 2], cam_intrinsic["intrinsic_M"][1, 2]
    return _warp_list(feat_img_src, d_candi, R, t, K, rays, cx, cy, align_corners)


def warp_img_feats_mgpu(feat_img_src, d_candi, R, t, IntM_tensors, unit_ray_arrays_2D, align_corners=False):
    """Same as warp_img_feats_v3 with the intrinsics passed as tensors ([1,3,3], [1,3,hw]);
    the principal point is read from the matrix (homography.py:416: `_back_warp_homo_parallel_v1`)."""
    K = IntM_tensors.squeeze(0).to(torch.float32)
    rays = unit_ray_arrays_2D.squeeze(0).to(torch.float32).contiguous()
    cx, cy = float(K[0, 2]), float(K[1, 2])
    return _warp_list(feat_img_src, d_candi, R, t, K, rays, cx, cy, align_corners)


def z_range(d_candi):
    """z_half / z_radius of the back-projected grid (homography.py:689-693); rays have z = 1."""
    d32 = np.asarray(d_candi).astype(np.float32)
    z_max, z_min = d32.max(), d32.min()
    return float(np.float32((z_max + z_min) * np.float32(0.5))), float(np.float32((z_max - z_min) * np.float32(0.5)))


def z_range_f64(d_candi):
    """The d_candi_new form (homography.py:685-693): z_max / z_min are numpy float64 scalars of d_candi, z_half / z_radius
    float64 products that torch casts to fp32 when they meet the fp32 coordinate tensor."""
    d = np.asarray(d_candi)
    z_max, z_min = d.max(), d.min()
    return float(np.float32((z_max + z_min) * .5)), float(np.float32((z_max - z_min) * .5))


def resample_vol_cuda(src_vol, rel_extM, cam_intrinsic=None, d_candi=None, d_candi_new=None,
                      padding_value=0., output_tensor=False, is_debug=False,
                      PointsDs_ref_cam_coord_in=None, clamp=None):
    """PREDICT step: resample the volume src_vol [1,D,h,w] under the rigid motion rel_extM [4,4].

    Returns [D,h,w].  d_candi_new (the LBA driver's form, test_KVNet_LBA.py:414-417): the planes are sampled at the new
    candidates while the depth axis is normalised by the float64 range of d_candi.  `clamp=(lo, hi)` fuses the `.clamp(min=lo, max=hi)` the callers apply
    (test_utils/test_KVNet.py:59); the default None matches the reference function itself.
    """
    assert d_candi is not None, 'd_candi should be some np.array object'
    if PointsDs_ref_cam_coord_in is not None or is_debug:
        raise NotImplementedError("a caller-supplied point grid / the debug return are not on this path "
                                  "(no reference script passes them)")
    D, h, w = src_vol.shape[1:]
    dev = src_vol.device
    _, rays = _cam_dev(cam_intrinsic, dev)
    hhfov = math.radians(cam_intrinsic['hfov']) * .5
    hvfov = math.radians(cam_intrinsic['vfov']) * .5
    T = rel_extM.to(device=dev, dtype=torch.float32)
    if d_candi_new is not None:
        # test_KVNet_LBA.py:414-417: sample at the NEW candidates, depth axis normalised by the source candidates' range
        # The reference allocates D (= source planes) point planes and fills the first len(d_candi_new) (:673-682): more
        # candidates than planes is its IndexError, fewer leaves the remaining planes at the origin (d = 0).
        z_half, z_radius = z_range_f64(d_candi)
        d_new = np.asarray(d_candi_new, dtype=np.float64).reshape(-1)
        if d_new.shape[0] > D:
            raise IndexError("index %d is out of bounds for dimension 1 with size %d" % (D, D))
        d_new = np.concatenate([d_new, np.zeros(D - d_new.shape[0])])
        return ops.dpv_resample(src_vol[0], T, rays, _d_candi_dev(d_new, dev), math.tan(hhfov), math.tan(hvfov),
                                z_half, z_radius, padding_value, clamp=clamp, new_candi=True)
    z_half, z_radius = z_range(d_candi)
    return ops.dpv_resample(src_vol[0], T, rays, _d_candi_dev(d_candi, dev), math.tan(hhfov),
                            math.tan(hvfov), z_half, z_radius, padding_value, clamp=clamp)


def back_warp_th_Rt_msrc(imgs_src, dmap, Rs, ts, cam_intrinsic):
    """Warp the N source frames imgs_src [N,C,H,W] to the reference view given the reference depth map dmap [H,W] and
    the motions Rs [N,3,3], ts [N,3] (reference -> source): [N,C,H,W].  Differentiable w.r.t. Rs and ts
    (the local bundle adjustment of ICP/opt_pose_numerical.py optimises them)."""
    assert isinstance(imgs_src, torch.Tensor)
    from .autograd import DepthWarp
    npts = dmap.numel()
    assert cam_intrinsic['unit_ray_array_2D'].shape[1] == npts
    dev = imgs_src.device
    K, rays = _cam_dev(cam_intrinsic, dev)
    dm = dmap.to(device=dev, dtype=torch.float32).reshape(imgs_src.shape[2], imgs_src.shape[3])
    Rs = Rs.to(device=dev, dtype=torch.float32).reshape(-1, 3, 3)
    ts = ts.to(device=dev, dtype=torch.float32).reshape(-1, 3)
    return DepthWarp.apply(imgs_src.to(torch.float32).contiguous(), dm.contiguous(), K, Rs, ts, rays)


def back_warp_th_Rt(img_src, dmap, R, t, cam_intrinsic):
    """Single-view form: img_src [1,C,H,W], R [3,3], t [3] -> [1,C,H,W]."""
    assert isinstance(R, torch.Tensor) and isinstance(t, torch.Tensor), 'R,t should be torch tensors'
    return back_warp_th_Rt_msrc(img_src, dmap, R.reshape(1, 3, 3), t.reshape(1, 3), cam_intrinsic)


def get_rel_extrinsicM(ext_ref, ext_src):
    """Extrinsic matrix from ref_view to src_view (numpy 4x4)."""
    return ext_src.dot(np.linalg.inv(ext_ref))
