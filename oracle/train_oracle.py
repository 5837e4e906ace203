"""CPU restatement of one training iteration of the depth path under torch autograd.

TEST INFRASTRUCTURE ONLY (tests/, bench.py's cpu_baseline leg of `--mode train`) — the product package never imports it.

Follows the reference's train() (train_utils/train_KVNet.py:20-203): KVNET.forward with gradients, NLL on the 1/4-resolution
volumes and on their R-Net refinements (:103-120), loss.backward() (:152), optimizer step, PREDICT on the detached volume
(:155-171).  The networks are the functional ones of kvnet_oracle (the same ATen CPU ops the reference calls).  The one thing
that cannot be reused is the C restatement of the sampling arithmetic — it has no gradient — so the cost volume is written the
way the reference itself computes it: homography.py:293-331 (P = K t + (K R ray) d, divide by z + 1e-10, normalise by the
principal point) + F.grid_sample(bilinear, zeros) per source view, and :421-448 (squared difference summed over channels / sigma).
The K-Net's warped colour planes carry no gradient (their inputs are images), so they stay on the C oracle.

Pinned by tests/test_oracle_golden.py against tests/golden/train_small.npz — two iterations of the unmodified reference's own
train() (oracle/gen_golden.py::gen_train): losses, predicted state, SGD weight deltas.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

from . import cpu_oracle as co
from . import kvnet_oracle as ko

CNN = "feature_extractor.feature_extraction"
TWIN = "d_net.feature_extraction.feature_extraction"     # the same module registered a second time (KVNET.py:72-76)


def costvol_autograd(feat_ref, feat_src, K, poses, rays, d_candi, sigma):
    """feat_ref [C,h,w], feat_src [V,C,h,w] -> cost [D,h,w], differentiable w.r.t. the features.
    homography.py:293-331 (back-projection + grid_sample) and :421-448 (L2 difference volume)."""
    V, C, h, w = feat_src.shape
    D = len(d_candi)
    d = torch.as_tensor(np.asarray(d_candi, np.float32)).reshape(D, 1, 1)
    cx, cy = float(K[0, 2]), float(K[1, 2])
    cost = torch.zeros(D, h, w)
    for v in range(V):
        R, t = poses[v, :3, :3], poses[v, :3, 3]
        term1 = torch.matmul(K, t).reshape(1, 3, 1)
        term2 = torch.matmul(torch.matmul(K, R), rays).unsqueeze(0)
        P = term1 + term2 * d
        P = P / (P[:, 2:3] + 1e-10)
        grid = torch.stack(((P[:, 0] - cx) / cx, (P[:, 1] - cy) / cy), -1).reshape(D, h, w, 2)
        warped = F.grid_sample(feat_src[v:v + 1].expand(D, C, h, w), grid, mode="bilinear", padding_mode="zeros",
                               align_corners=False)
        cost = cost + (warped - feat_ref.unsqueeze(0)).pow(2).sum(1) / sigma
    return cost


def forward_autograd(sd, ref, src, poses, cam, d_candi, sigma, BV_predict=None):
    """kvnet_oracle.kvnet_forward with a gradient path: (R(BV_cur), R(DPV), BV_cur, DPV), KVNET.py:93-185."""
    frames = torch.cat((src[0], ref), 0)
    layer1, feats = ko.feature_cnn(sd, CNN, frames)
    dw = int(ref.shape[3] / feats.shape[3])
    full = torch.cat((feats, F.avg_pool2d(frames, dw)), 1)                     # basic.py:254-263
    K, rays = cam["intrinsic_M_cuda"], cam["unit_ray_array_2D"]
    cost = costvol_autograd(full[-1], full[:-1], K, poses[0], rays, d_candi, sigma)
    BV_cur = F.log_softmax(-cost, dim=0)[None]                                 # basic.py:299-300
    rfeats = [feats[-1:], layer1[-1:], ref]
    R_cur = ko.rnet(sd, "r_net", torch.exp(BV_cur), rfeats)
    if BV_predict is None or bool(torch.isnan(BV_predict[0, 0, 0, 0])):
        return R_cur, R_cur, BV_cur, BV_cur
    V = src.shape[1]
    rgb = full[:, -3:].detach()
    KR, Kt = ko._terms(cam, poses[0])
    warped = co.warp_volume(rgb[:V].numpy(), KR, Kt, rays.numpy(), d_candi, cam["intrinsic_M"][0, 2], cam["intrinsic_M"][1, 2])
    D = len(d_candi)
    h, w = rgb.shape[2:]
    vol = torch.cat((torch.from_numpy(warped).reshape(V * 3, D, h, w), rgb[V][:, None].expand(3, D, h, w),
                     (BV_cur - BV_predict)), 0)[None]                          # KVNET.py:163-166
    gain = ko.knet(sd, "kv_net", vol)
    DPV = F.log_softmax(BV_predict + gain[0], dim=1)                           # KVNET.py:172-173
    R_kv = ko.rnet(sd, "r_net", torch.exp(DPV), rfeats)
    return R_cur, R_kv, BV_cur, DPV


def leaf_state(sd):
    """State dict -> the same keys as leaf tensors that require a gradient (floating-point parameters only; the twin
    registration of the feature CNN shares its tensors, as the reference's modules do)."""
    out = {}
    for k, v in sd.items():
        if k.startswith(TWIN):
            continue
        t = v.detach().clone().float()
        if "running_" not in k and "num_batches" not in k and t.is_floating_point():
            t.requires_grad_(True)
        out[k] = t
    for k in list(out):
        if k.startswith(CNN):
            out[TWIN + k[len(CNN):]] = out[k]
    return out


def parameters(leaves):
    seen, ps = set(), []
    for k in sorted(leaves):
        t = leaves[k]
        if t.requires_grad and id(t) not in seen:
            seen.add(id(t))
            ps.append(t)
    return ps


def train_iteration(leaves, opt, ref, src, poses, dmap, dmap_full, cam, d_candi, sigma, BV_predict, t_win_r=2):
    """One call of the reference's train(): -> (loss, BV_predict for the next window).  `opt` is a torch optimizer over
    parameters(leaves)."""
    valid = isinstance(BV_predict, torch.Tensor) and not bool(torch.isnan(BV_predict[0, 0, 0, 0]))
    opt.zero_grad()
    R_cur, R_kv, BV_cur, DPV = forward_autograd(leaves, ref, src, poses, cam, d_candi, sigma, BV_predict if valid else None)
    loss = F.nll_loss(BV_cur, dmap, ignore_index=0) + F.nll_loss(R_cur, dmap_full, ignore_index=0)
    if valid:
        loss = loss + F.nll_loss(DPV, dmap, ignore_index=0) + F.nll_loss(R_kv, dmap_full, ignore_index=0)
    loss.backward()
    opt.step()
    with torch.no_grad():
        nxt = ko.predict(DPV.detach(), poses[0, t_win_r], cam, d_candi)
        nxt = nxt.clamp(-1000., 0.)
    return loss.detach(), nxt
