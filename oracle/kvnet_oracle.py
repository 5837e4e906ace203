"""CPU restatement of the whole depth path: KVNET.forward + the PREDICT step.

TEST INFRASTRUCTURE ONLY (tests/, __graft_entry__.smoke(), bench.py cpu_baseline) — the product
package never imports it.

Functional (no nn.Module), driven by a state dict with the reference's parameter names.  The
sampling arithmetic is the C restatement (oracle/nrgbd_oracle.c via cpu_oracle); convolutions,
batch-norm, pooling and interpolation are the same ATen CPU ops the reference calls (their
arithmetic lives in PyTorch, not in /root/reference — SURVEY.md §8c "third-party arithmetic").
Pinned against the unmodified reference by tests/test_oracle_vs_reference.py (runs where
/root/reference exists) and by the golden vectors under tests/golden/.

Every BatchNorm uses batch statistics: the reference never calls .eval(), and in
train() mode running statistics do not enter the output.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

from . import cpu_oracle as co


# ----------------------------------------------------------------------------- building blocks
def _bn(x, sd, p):
    return F.batch_norm(x, None, None, sd[p + ".weight"], sd[p + ".bias"], training=True, eps=1e-5)


def _convbn2d(x, sd, p, stride=1, pad=1, dil=1):
    """psm_submodule.py:10-16 convbn: padding = dilation if dilation > 1 else pad."""
    x = F.conv2d(x, sd[p + ".0.weight"], None, stride, dil if dil > 1 else pad, dil)
    return _bn(x, sd, p + ".1")


def _block2d(x, sd, p, stride, pad, dil):
    """psm_submodule.py:31-50 BasicBlock."""
    y = F.relu(_convbn2d(x, sd, p + ".conv1.0", stride, pad, dil))
    y = _convbn2d(y, sd, p + ".conv2", 1, pad, dil)
    if (p + ".downsample.0.weight") in sd:
        x = _bn(F.conv2d(x, sd[p + ".downsample.0.weight"], None, stride), sd, p + ".downsample.1")
    return y + x


def feature_cnn(sd, p, x):
    """psm_submodule.py:141-167 feature_extraction.forward -> (layer1, feature)."""
    for i in (0, 2, 4):
        x = F.relu(_convbn2d(x, sd, "%s.firstconv.%d" % (p, i), 2 if i == 0 else 1, 1, 1))
    for b in range(3):
        x = _block2d(x, sd, "%s.layer1.%d" % (p, b), 1, 1, 1)
    layer1 = x
    for b in range(16):
        x = _block2d(x, sd, "%s.layer2.%d" % (p, b), 2 if b == 0 else 1, 1, 1)
    raw = x
    for b in range(3):
        x = _block2d(x, sd, "%s.layer3.%d" % (p, b), 1, 1, 1)
    for b in range(3):
        x = _block2d(x, sd, "%s.layer4.%d" % (p, b), 1, 1, 2)
    skip = x
    size = skip.shape[2:]
    branches = {}
    for i, win in zip((1, 2, 3, 4), (64, 32, 16, 8)):
        y = F.avg_pool2d(skip, (win, win), stride=(win, win))
        y = F.relu(_convbn2d(y, sd, "%s.branch%d.1" % (p, i), 1, 0, 1))
        branches[i] = F.interpolate(y, size=size, mode="bilinear", align_corners=True)
    x = torch.cat((raw, skip, branches[4], branches[3], branches[2], branches[1]), 1)
    x = F.relu(_convbn2d(x, sd, p + ".lastconv.0", 1, 1, 1))
    x = F.conv2d(x, sd[p + ".lastconv.2.weight"])
    return layer1, x


def knet(sd, p, vol):
    """basic.py:113-139 KV_NET_BASIC.forward (if_normalize False, no up-sampling)."""
    def cb(x, q):
        return _bn(F.conv3d(x, sd[q + ".0.weight"], None, 1, 1), sd, q + ".1")
    x = F.relu(cb(vol, p + ".dres0.0"))
    x = F.relu(cb(x, p + ".dres0.2"))
    for i in (1, 2, 3, 4):
        y = F.relu(cb(x, "%s.dres%d.0" % (p, i)))
        x = cb(y, "%s.dres%d.2" % (p, i)) + x
    y = F.relu(cb(x, p + ".classify.0"))
    return F.conv3d(y, sd[p + ".classify.2.weight"], None, 1, 1)


def rnet(sd, p, dpv, feats):
    """Refine.py:79-107 RefineNet_DPV_upsample.forward."""
    def cl(x, q):
        return F.leaky_relu(F.conv2d(x, sd[q + ".0.weight"], sd[q + ".0.bias"], 1, 1), 0.01)

    def tl(x, q):
        return F.leaky_relu(F.conv_transpose2d(x, sd[q + ".0.weight"], sd[q + ".0.bias"], 2, 1), 0.01)
    x = cl(cl(torch.cat([dpv, feats[0]], 1), p + ".conv0"), p + ".conv0_1")
    x = tl(x, p + ".trans_conv0")
    x = cl(cl(torch.cat([x, feats[1]], 1), p + ".conv1"), p + ".conv1_1")
    x = tl(x, p + ".trans_conv1")
    x = cl(cl(torch.cat([x, feats[2]], 1), p + ".conv2"), p + ".conv2_1")
    x = F.conv2d(x, sd[p + ".conv2_2.weight"], sd[p + ".conv2_2.bias"], 1, 1)
    return F.log_softmax(x, dim=1)


# ----------------------------------------------------------------------------- geometry
def _terms(cam, poses):
    """homography.py:315-317 (term1 = K t, left factor K R of term2) in the reference's CPU summation order."""
    return co.homography_terms(cam["intrinsic_M_cuda"].numpy(), poses[:, :3, :3].numpy(), poses[:, :3, 3].numpy())


def dnet(sd, ref, src, poses, cam, d_candi, sigma, feat_dist="L2"):
    """basic.py:223-323 D_NET_BASIC.forward (use_img_intensity, BV_log, output_features)."""
    frames = torch.cat((src[0], ref), 0)
    layer1, feats = feature_cnn(sd, "feature_extractor.feature_extraction", frames)
    dw = int(ref.shape[3] / feats.shape[3])
    full = torch.cat((feats, F.avg_pool2d(frames, dw)), 1)                 # basic.py:254-263
    KR, Kt = _terms(cam, poses[0])
    cost = co.costvol(full[-1].numpy(), full[:-1].numpy(), KR, Kt, cam["unit_ray_array_2D"].numpy(),
                      d_candi, cam["intrinsic_M"][0, 2], cam["intrinsic_M"][1, 2], sigma, dist=feat_dist)
    BV = torch.from_numpy(co.logsoftmax_d(cost, scale=-1.0))[None]         # basic.py:299-300
    return BV, [feats[-1:], layer1[-1:]], full


def kvnet_forward(sd, ref, src, poses, cam, d_candi, sigma, BV_predict=None):
    """KVNET.py:93-185 -> (R(BV_cur), R(DPV), BV_cur, DPV); first frame returns the D-Net pair twice."""
    BV_cur, feats, full = dnet(sd, ref, src, poses, cam, d_candi, sigma)
    feats = feats + [ref]
    R_cur = rnet(sd, "r_net", torch.exp(BV_cur), feats)
    if BV_predict is None or bool(torch.isnan(BV_predict[0, 0, 0, 0])):
        return R_cur, R_cur, BV_cur, BV_cur
    V = src.shape[1]
    rgb = full[:, -3:]                                                     # KVNET.py:149-151 (same pooling)
    KR, Kt = _terms(cam, poses[0])
    warped = co.warp_volume(rgb[:V].numpy(), KR, Kt, cam["unit_ray_array_2D"].numpy(), d_candi,
                            cam["intrinsic_M"][0, 2], cam["intrinsic_M"][1, 2])      # [V,3,D,h,w]
    D = len(d_candi)
    h, w = rgb.shape[2:]
    ref_rep = rgb[V][:, None].expand(3, D, h, w)
    vol = torch.cat((torch.from_numpy(warped).reshape(V * 3, D, h, w), ref_rep,
                     (BV_cur - BV_predict)), 0)[None]                      # KVNET.py:163-166
    gain = knet(sd, "kv_net", vol)
    DPV = torch.from_numpy(co.logsoftmax_d(gain[0, 0].numpy(), BV_predict[0].numpy()))[None]  # :172-173
    R_kv = rnet(sd, "r_net", torch.exp(DPV), feats)
    return R_cur, R_kv, BV_cur, DPV


def predict(dpv, pose_next, cam, d_candi, rel_extM=None):
    """test_utils/test_KVNet.py:47-62: resample by inverse(pose ref->next), pad log(1/D), clamp [-1000,0].
    The inverse is the path's fixed-order one (oracle_pose_inverse == nrgbd_pose_inverse); `rel_extM` substitutes a given
    matrix — the golden tests pass the reference's own host-LAPACK result (tests/golden/pose_inv_ref.npz) to show that the
    last bits of that matrix are the ONLY difference between this PREDICT and the reference's."""
    T = co.pose_inverse(pose_next.numpy()) if rel_extM is None else np.asarray(rel_extM, np.float32)
    D = len(d_candi)
    out = co.dpv_resample(dpv[0].numpy(), T, cam["unit_ray_array_2D"].numpy(), d_candi,
                          math.tan(math.radians(cam["hfov"]) * .5), math.tan(math.radians(cam["vfov"]) * .5),
                          math.log(1. / float(D)))
    return torch.from_numpy(out)[None]


def step_full(sd, ref, src, poses, cam, d_candi, sigma, BV_predict, t_win_r=2, pose_next=None, rel_extM=None):
    """step() keeping BOTH refined outputs: (R_cur, R_kv, DPV, BV_cur, BV_predict_next) — R_cur = R-Net(BV_cur), R_kv = R-Net(DPV)
    (KVNET.py:128,176); on the first frame DPV = BV_cur and R_kv = R_cur."""
    with torch.no_grad():
        R_cur, R_kv, BV_cur, DPV = kvnet_forward(sd, ref, src, poses, cam, d_candi, sigma, BV_predict)
        nxt = predict(DPV, poses[0, t_win_r] if pose_next is None else pose_next, cam, d_candi, rel_extM)
    return R_cur, R_kv, DPV, BV_cur, nxt


def step(sd, ref, src, poses, cam, d_candi, sigma, BV_predict, t_win_r=2, pose_next=None, rel_extM=None):
    """One iteration of the reference's test(): forward + PREDICT -> (R_kv, DPV, BV_cur, BV_predict_next)."""
    with torch.no_grad():
        R_cur, R_kv, BV_cur, DPV = kvnet_forward(sd, ref, src, poses, cam, d_candi, sigma, BV_predict)
        nxt = predict(DPV, poses[0, t_win_r] if pose_next is None else pose_next, cam, d_candi, rel_extM)
    return R_kv, DPV, BV_cur, nxt
