"""Camera-intrinsics dict of the plane-sweep path (host side, numpy only).

Mirrors the schema the reference builds in code/mdataloader/scanNet.py:204-272
(read_IntM_from_txt) with the per-pixel ray table of code/warping/View.py:16-62:

    {'hfov', 'vfov'              degrees
     'unit_ray_array'            (h, w, 3) float64, z = 1 rays
     'unit_ray_array_2D'         (3, h*w) float32 torch tensor
     'intrinsic_M_cuda'          (3, 3) float32 torch tensor
     'intrinsic_M'               (3, 4) float64
     'focal_length'}

The reference fills the ray table with a Python double loop over h*w pixels; here the same
IEEE-754 expression is evaluated with numpy broadcasting, which is bit-identical (the only
libm call, tan, stays `math.tan` on a scalar).
"""
import math

import numpy as np
import torch

# code/DSO/cam_info_scanNet.mat: IntM = [[1169.62, 0, 646.295], [0, 1167.11, 489.927]], 1296x968
SCANNET_K = ((1169.62, 0.0, 646.295), (0.0, 1167.11, 489.927), (0.0, 0.0, 1.0))
# code/DSO/cam_info_kitti.mat
KITTI_K = ((735.0801008203796, 0.0, 621.0), (0.0, 782.6739256829462, 187.5), (0.0, 0.0, 1.0))
# code/DSO/cam_info_7scenes.mat
SEVEN_SCENES_K = ((585.0, 0.0, 320.0), (0.0, 585.0, 240.0), (0.0, 0.0, 1.0))


def fov_from_K(K):
    """scanNet.py:239-240: fov = 2*atan(c / f) in degrees."""
    h_fov = math.degrees(math.atan(K[0][2] / K[0][0]) * 2)
    v_fov = math.degrees(math.atan(K[1][2] / K[1][1]) * 2)
    return h_fov, v_fov


def unit_ray_array(width, height, hfov, vfov):
    """View.py:16-62 with normalize_z=True: ray = (tan(hfov/2)(2(x+.5)/W - 1), tan(vfov/2)(2(y+.5)/H - 1), 1)."""
    tx = math.tan(math.radians(hfov / 2.0))
    ty = math.tan(math.radians(vfov / 2.0))
    xs = np.arange(width, dtype=np.float64)
    ys = np.arange(height, dtype=np.float64)
    x_vect = tx * ((2.0 * ((xs + 0.5) / width)) - 1.0)
    y_vect = ty * ((2.0 * ((ys + 0.5) / height)) - 1.0)
    rays = np.empty((height, width, 3), dtype=np.float64)
    rays[:, :, 0] = x_vect[None, :]
    rays[:, :, 1] = y_vect[:, None]
    rays[:, :, 2] = 1.0
    return rays


def make_cam_intrinsics(hfov, vfov, width, height, focal_length=None):
    """Intrinsics rebuilt at the plane-sweep grid size (scanNet.py:243-270, the out_size branch)."""
    K = np.zeros((3, 4))
    K[2, 2] = 1.0
    K[0, 0] = (width / 2.0) / math.tan(math.radians(hfov / 2.0))
    K[0, 2] = width / 2.0
    K[1, 1] = (height / 2.0) / math.tan(math.radians(vfov / 2.0))
    K[1, 2] = height / 2.0
    rays = unit_ray_array(width, height, hfov, vfov)
    rays_2d = np.reshape(np.transpose(rays, axes=[2, 0, 1]), [3, -1])
    return {
        "hfov": hfov,
        "vfov": vfov,
        "unit_ray_array": rays,
        "unit_ray_array_2D": torch.from_numpy(rays_2d.astype(np.float32)),
        "intrinsic_M_cuda": torch.from_numpy(K[:3, :3].astype(np.float32)),
        "focal_length": float(np.mean([K[0, 0], K[1, 1]])) if focal_length is None else focal_length,
        "intrinsic_M": K,
    }


def cam_intrinsics_from_K(K, width, height):
    """Dict for a grid of width x height from a full-size pinhole matrix (FOV is size-invariant)."""
    hfov, vfov = fov_from_K(K)
    return make_cam_intrinsics(hfov, vfov, width, height)


def scannet_intrinsics(width, height):
    return cam_intrinsics_from_K(SCANNET_K, width, height)


def kitti_intrinsics(width, height):
    return cam_intrinsics_from_K(KITTI_K, width, height)
