"""float64 evaluation of the SAME graph as oracle/kvnet_oracle.py — the yardstick for fp32 rounding noise.

TEST INFRASTRUCTURE ONLY (tests/, oracle/gen_golden.py).

The contract asks for "DPV floats within 1e-4" of the reference.  The reference itself is an fp32 program whose
result depends on the summation order of ATen's CPU kernels, so |GPU - reference| mixes two rounding errors.  This
module evaluates the identical formulas (models/KVNET.py:93-185, warping/homography.py:293-331,421-448,654-723,
test_utils/test_KVNet.py:47-62) in float64 with the fp32 inputs (weights, images, poses, intrinsics, ray table,
d_candi cast to fp32 as homography.py:311 does) taken as exact, so that |x - fp64| can be reported separately for
x = the GPU path, x = the CPU oracle and x = the reference (the latter stored in tests/golden/ by gen_golden.py).
Sampling goes through F.grid_sample on float64 tensors, exactly the call the reference makes.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

from . import kvnet_oracle as ko

F64 = torch.float64


def _cam64(cam):
    K = cam["intrinsic_M_cuda"].to(F64)
    rays = cam["unit_ray_array_2D"].to(F64)
    return K, rays


def _sweep_grid(K, rays, R, t, d, cx, cy, h, w):
    """homography.py:315-317,433-445 for one view: -> grid [D,h,w,2]."""
    term1 = K.matmul(t).reshape(3, 1)
    term2 = K.matmul(R).matmul(rays)
    P = term1.unsqueeze(0) + term2.unsqueeze(0) * d.reshape(-1, 1, 1)
    P = P / (P[:, 2, :].unsqueeze(1) + 1e-10)
    gx = (P[:, 0, :] - cx) / cx
    gy = (P[:, 1, :] - cy) / cy
    return torch.stack((gx, gy), -1).reshape(-1, h, w, 2)


def costvol(full, poses, cam, d_candi, sigma):
    """est_swp_volume_v4 (L2) in float64: full [V+1,C,h,w] (last = reference) -> cost [D,h,w]."""
    K, rays = _cam64(cam)
    V = full.shape[0] - 1
    h, w = full.shape[2:]
    d = torch.from_numpy(np.asarray(d_candi).astype(np.float32)).to(F64)
    cx, cy = float(cam["intrinsic_M"][0, 2]), float(cam["intrinsic_M"][1, 2])
    cost = torch.zeros(len(d), h, w, dtype=F64)
    for v in range(V):
        grid = _sweep_grid(K, rays, poses[v, :3, :3].to(F64), poses[v, :3, 3].to(F64), d, cx, cy, h, w)
        warped = F.grid_sample(full[v:v + 1].expand(len(d), -1, -1, -1), grid, mode="bilinear", padding_mode="zeros",
                               align_corners=False)
        cost = cost + ((warped - full[V:V + 1]) ** 2).sum(1) / sigma
    return cost


def warp_rgb(rgb_src, poses, cam, d_candi):
    """warp_img_feats_v3 in float64: rgb_src [V,3,h,w] -> [V*3, D, h, w] (KVNET.py:158-166 order)."""
    K, rays = _cam64(cam)
    V, _, h, w = rgb_src.shape
    d = torch.from_numpy(np.asarray(d_candi).astype(np.float32)).to(F64)
    cx, cy = float(cam["intrinsic_M"][0, 2]), float(cam["intrinsic_M"][1, 2])
    out = []
    for v in range(V):
        grid = _sweep_grid(K, rays, poses[v, :3, :3].to(F64), poses[v, :3, 3].to(F64), d, cx, cy, h, w)
        warped = F.grid_sample(rgb_src[v:v + 1].expand(len(d), -1, -1, -1), grid, mode="bilinear", padding_mode="zeros",
                               align_corners=False)
        out.append(warped.transpose(0, 1))   # [3,D,h,w]
    return torch.cat(out, 0)


def predict(dpv, pose_next, cam, d_candi):
    """resample_vol_cuda + _set_vol_border + clamp in float64 (homography.py:654-723,873-887)."""
    D, h, w = dpv.shape[1:]
    _, rays = _cam64(cam)
    T = torch.linalg.inv(pose_next.to(F64))
    d = torch.from_numpy(np.asarray(d_candi).astype(np.float32)).to(F64)
    X = d.reshape(D, 1, 1) * rays.reshape(1, 3, h * w)              # [D,3,hw]
    Xs = T[:3, :3].matmul(X) + T[:3, 3].reshape(1, 3, 1)
    z_max, z_min = d.max(), d.min()
    z_half, z_rad = (z_max + z_min) * .5, (z_max - z_min) * .5
    gx = Xs[:, 0] / (Xs[:, 2] + 1e-10) / math.tan(math.radians(cam["hfov"]) * .5)
    gy = Xs[:, 1] / (Xs[:, 2] + 1e-10) / math.tan(math.radians(cam["vfov"]) * .5)
    gz = (Xs[:, 2] - z_half) / z_rad
    grid = torch.stack((gx, gy, gz), -1).reshape(1, D, h, w, 3)
    vol = dpv.clone().reshape(1, 1, D, h, w)
    pad = math.log(1. / float(D))
    vol[:, :, 0] = pad; vol[:, :, -1] = pad
    vol[:, :, :, 0] = pad; vol[:, :, :, -1] = pad
    vol[:, :, :, :, 0] = pad; vol[:, :, :, :, -1] = pad
    out = F.grid_sample(vol, grid, mode="bilinear", padding_mode="border", align_corners=False)[0, 0]
    return out.clamp(min=-1000., max=0.)[None]


def step(sd, ref, src, poses, cam, d_candi, sigma, BV_predict, t_win_r=2):
    """One iteration of test(): (R_kv, DPV, BV_cur, BV_predict_next), everything float64."""
    sd = {k: (v.to(F64) if v.is_floating_point() else v) for k, v in sd.items()}
    ref, src, poses = ref.to(F64), src.to(F64), poses.to(F64)
    with torch.no_grad():
        frames = torch.cat((src[0], ref), 0)
        layer1, feats = ko.feature_cnn(sd, "feature_extractor.feature_extraction", frames)
        dw = int(ref.shape[3] / feats.shape[3])
        full = torch.cat((feats, F.avg_pool2d(frames, dw)), 1)
        BV_cur = torch.log_softmax(-costvol(full, poses[0], cam, d_candi, sigma), dim=0)[None]
        fl = [feats[-1:], layer1[-1:], ref]
        R_cur = ko.rnet(sd, "r_net", torch.exp(BV_cur), fl)
        if BV_predict is None:
            R_kv, DPV = R_cur, BV_cur
        else:
            V = src.shape[1]
            rgb = full[:, -3:]
            D = len(d_candi)
            h, w = rgb.shape[2:]
            vol = torch.cat((warp_rgb(rgb[:V], poses[0], cam, d_candi), rgb[V][:, None].expand(3, D, h, w),
                             BV_cur - BV_predict.to(F64)), 0)[None]
            gain = ko.knet(sd, "kv_net", vol)
            DPV = torch.log_softmax(gain[:, 0] + BV_predict.to(F64), dim=1)
            R_kv = ko.rnet(sd, "r_net", torch.exp(DPV), fl)
        nxt = predict(DPV, poses[0, t_win_r], cam, d_candi)
    return R_kv, DPV, BV_cur, nxt
