"""Seeded synthetic windows, poses and weights for the plane-sweep depth path.

There is no dataset and no checkpoint offline, so the benchmark, the parity tests and the
golden-vector generator all draw their inputs from here: images N(0,1)
fp32 (the reference loader emits ImageNet-normalised RGB), poses = small rigid motions
(rotation-vector sigma 0.02 rad, translation sigma 0.05 m), ScanNet field of view.
Everything is numpy `RandomState` driven, hence identical on every machine.
"""
import zlib

import numpy as np
import torch


def rotvec_to_R(rv):
    """Rodrigues formula, float64."""
    rv = np.asarray(rv, np.float64)
    th = np.linalg.norm(rv)
    if th < 1e-12:
        return np.eye(3)
    k = rv / th
    Kx = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    return np.eye(3) + np.sin(th) * Kx + (1 - np.cos(th)) * (Kx @ Kx)


def random_pose(rng, rot_sigma=0.02, trans_sigma=0.05):
    """4x4 rigid transform ref -> src (the convention of homography.py:904 get_rel_extrinsicM)."""
    T = np.eye(4)
    T[:3, :3] = rotvec_to_R(rng.normal(0, rot_sigma, 3))
    T[:3, 3] = rng.normal(0, trans_sigma, 3)
    return T


def random_poses(rng, V, rot_sigma=0.02, trans_sigma=0.05):
    return np.stack([random_pose(rng, rot_sigma, trans_sigma) for _ in range(V)]).astype(np.float32)


def noise_window(seed, H, W, V=4):
    """(ref [1,3,H,W], src [1,V,3,H,W], poses [1,V,4,4]) — the pure-noise input family."""
    rng = np.random.RandomState(seed)
    ref = rng.standard_normal((1, 3, H, W)).astype(np.float32)
    src = rng.standard_normal((1, V, 3, H, W)).astype(np.float32)
    poses = random_poses(rng, V)[None]
    return torch.from_numpy(ref), torch.from_numpy(src), torch.from_numpy(poses)


def smooth_texture(rng, C, H, W, octaves=4):
    """Band-limited random texture (sum of upsampled noise octaves), fp32 [C,H,W], ~unit variance."""
    out = np.zeros((C, H, W), np.float64)
    for o in range(octaves):
        s = 2 ** (o + 1)
        hh, ww = max(2, H // (32 // min(s, 32)) // 2), max(2, W // (32 // min(s, 32)) // 2)
        g = torch.from_numpy(rng.standard_normal((1, C, hh, ww)))
        up = torch.nn.functional.interpolate(g, size=(H, W), mode="bicubic", align_corners=False)
        out += up[0].numpy() / (o + 1)
    out /= out.std() + 1e-12
    return out.astype(np.float32)


def rendered_window(seed, H, W, cam_full, V=4, d_lo=0.6, d_hi=4.0):
    """Window whose sources are renderings of a textured scene, so the cost volume has a true minimum.

    The scene is a smooth random depth map in the reference view with a random texture.  Each
    source view is produced by forward-projecting a dense source-pixel grid into the reference
    through the *reference* depth (fixed-point iteration on the source depth), i.e. by the
    inverse of the warp the plane sweep undoes.  Returns tensors like `noise_window` plus the
    reference depth map [H,W] (float32).
    """
    rng = np.random.RandomState(seed)
    tex = smooth_texture(rng, 3, H, W)
    z = smooth_texture(rng, 1, H, W, octaves=2)[0]
    z = (z - z.min()) / (z.max() - z.min() + 1e-12)
    depth = (d_lo + (d_hi - d_lo) * z).astype(np.float32)
    poses = random_poses(rng, V)
    K = cam_full["intrinsic_M"][:3, :3]
    Kinv = np.linalg.inv(K)
    ys, xs = np.meshgrid(np.arange(H) + 0.5, np.arange(W) + 0.5, indexing="ij")
    pix = np.stack([xs, ys, np.ones_like(xs)], 0).reshape(3, -1)
    tex_t = torch.from_numpy(tex)[None]
    depth_t = torch.from_numpy(depth)[None, None]
    srcs = []
    for v in range(V):
        T = poses[v].astype(np.float64)
        Tinv = np.linalg.inv(T)
        rays_s = Kinv @ pix
        ds = np.full(pix.shape[1], 0.5 * (d_lo + d_hi))
        for _ in range(8):  # fixed point: depth of the source pixel such that it lands on the ref surface
            Xs = rays_s * ds
            Xr = Tinv[:3, :3] @ Xs + Tinv[:3, 3:4]
            ur = K @ (Xr / Xr[2:3])
            g = np.stack([ur[0] / (W / 2.0) - 1.0, ur[1] / (H / 2.0) - 1.0], -1).reshape(1, H, W, 2)
            zr = torch.nn.functional.grid_sample(depth_t.double(), torch.from_numpy(g), mode="bilinear",
                                                 padding_mode="border", align_corners=False)[0, 0].numpy().reshape(-1)
            ds = ds * (zr / np.maximum(Xr[2], 1e-6))
        img = torch.nn.functional.grid_sample(tex_t.double(), torch.from_numpy(g), mode="bilinear",
                                              padding_mode="border", align_corners=False)[0]
        srcs.append(img.float())
    ref = torch.from_numpy(tex)[None]
    src = torch.stack(srcs)[None]
    return ref, src, torch.from_numpy(poses[None]), depth


def _render_view(tex_t, depth_t, K, Kinv, T, H, W, d_lo, d_hi):
    """The scene (texture + depth map in the scene view) seen from the camera X_view = T X_scene: forward projection of a dense
    pixel grid of the new view through the scene depth (fixed point on the new view's depth), as rendered_window does."""
    ys, xs = np.meshgrid(np.arange(H) + 0.5, np.arange(W) + 0.5, indexing="ij")
    pix = np.stack([xs, ys, np.ones_like(xs)], 0).reshape(3, -1)
    Tinv = np.linalg.inv(T.astype(np.float64))
    rays_s = Kinv @ pix
    ds = np.full(pix.shape[1], 0.5 * (d_lo + d_hi))
    g = None
    for _ in range(8):
        Xs = rays_s * ds
        Xr = Tinv[:3, :3] @ Xs + Tinv[:3, 3:4]
        ur = K @ (Xr / Xr[2:3])
        g = np.stack([ur[0] / (W / 2.0) - 1.0, ur[1] / (H / 2.0) - 1.0], -1).reshape(1, H, W, 2)
        zr = torch.nn.functional.grid_sample(depth_t.double(), torch.from_numpy(g), mode="bilinear",
                                             padding_mode="border", align_corners=False)[0, 0].numpy().reshape(-1)
        ds = ds * (zr / np.maximum(Xr[2], 1e-6))
    img = torch.nn.functional.grid_sample(tex_t.double(), torch.from_numpy(g), mode="bilinear", padding_mode="border",
                                          align_corners=False)[0]
    return img.float()


def rendered_stream(seed, H, W, cam_full, n_frames=2, V=4, d_lo=0.6, d_hi=4.0):
    """A short VIDEO of one textured scene: n_frames consecutive windows (ref, src, poses) whose images are all renderings of
    the same scene, with the reference camera moving the way the reference's driver loop assumes — the next reference frame
    is this window's source t_win_r = 2 (test_KVNet.py:47-62 predicts the DPV into src_cam_poses[:, t_win_r]).  The cost
    volume then has a true minimum, the DPV is peaked (log-probabilities of -50 and below away from the surface), and the
    PREDICT step carries a consistent belief into the next frame: the regime the filter runs in, unlike N(0,1) windows.
    Returns [(ref [1,3,H,W], src [1,V,3,H,W], poses [1,V,4,4]), ...]."""
    rng = np.random.RandomState(seed)
    tex = smooth_texture(rng, 3, H, W)
    z = smooth_texture(rng, 1, H, W, octaves=2)[0]
    z = (z - z.min()) / (z.max() - z.min() + 1e-12)
    depth = (d_lo + (d_hi - d_lo) * z).astype(np.float32)
    K = cam_full["intrinsic_M"][:3, :3]
    Kinv = np.linalg.inv(K)
    tex_t, depth_t = torch.from_numpy(tex)[None], torch.from_numpy(depth)[None, None]
    A = np.eye(4)                                   # scene -> current reference camera
    ref = tex_t.clone()
    out = []
    for _ in range(n_frames):
        rel = random_poses(rng, V)                  # reference -> source v
        views = [_render_view(tex_t, depth_t, K, Kinv, rel[v].astype(np.float64) @ A, H, W, d_lo, d_hi) for v in range(V)]
        out.append((ref.float().clone(), torch.stack(views)[None], torch.from_numpy(rel[None])))
        A = rel[2].astype(np.float64) @ A           # the next reference camera = this window's source 2
        ref = views[2][None]
    return out


def seeded_state_dict(model, seed=0):
    """Deterministic, name-keyed weights for any module with the KVNET parameter names.

    Keyed by parameter NAME (crc32) rather than construction order, so the reference model and
    this package's model receive identical tensors.  Distributions follow the reference
    initialisers (models/basic.py:29-43,97-111; models/Refine.py:109-132): conv weights
    N(0, sqrt(2/(k..k*out))); BatchNorm gamma ~ 1, beta ~ 0 (with a small seeded spread so the
    affine terms are exercised); transposed convs = bilinear kernel plus small noise; biases small.
    """
    out = {}
    state = model.state_dict()
    # the feature CNN is registered twice (feature_extractor.* and d_net.feature_extraction.*, same
    # storage): aliases must receive the same values -> key the generator by the smallest alias
    canon = {}
    for name, ref in state.items():
        key = (ref.data_ptr(), tuple(ref.shape))
        canon[key] = min(canon.get(key, name), name)
    for name, ref in state.items():
        cname = canon[(ref.data_ptr(), tuple(ref.shape))]
        rng = np.random.RandomState((zlib.crc32(cname.encode()) + 7919 * seed) & 0x7FFFFFFF)
        shape = tuple(ref.shape)
        if name.endswith("num_batches_tracked"):
            out[name] = torch.zeros((), dtype=ref.dtype)
        elif name.endswith("running_mean"):
            out[name] = torch.zeros(shape)
        elif name.endswith("running_var"):
            out[name] = torch.ones(shape)
        elif ref.dim() >= 4 and "trans_conv" in name and name.endswith("weight"):
            n = shape[-1]
            factor = (n + 1) // 2
            center = factor - 1 if n % 2 == 1 else factor - 0.5
            og = np.ogrid[:n, :n]
            bil = (1 - abs(og[0] - center) / factor) * (1 - abs(og[1] - center) / factor)
            w = np.broadcast_to(bil, shape) / shape[0] + rng.normal(0, 0.02 / shape[0], shape)
            out[name] = torch.from_numpy(w.astype(np.float32))
        elif ref.dim() >= 4:  # conv2d [O,I,k,k] / conv3d [O,I,k,k,k]
            n = int(np.prod(shape[2:])) * shape[0]
            out[name] = torch.from_numpy(rng.normal(0, np.sqrt(2.0 / n), shape).astype(np.float32))
        elif name.endswith("weight"):  # BatchNorm gamma
            out[name] = torch.from_numpy((1.0 + 0.1 * rng.standard_normal(shape)).astype(np.float32))
        else:  # BatchNorm beta / conv bias
            out[name] = torch.from_numpy((0.05 * rng.standard_normal(shape)).astype(np.float32))
    return out


def trained_like_state_dict(model, seed=0, offset=2.0):
    """Weights in the regime of a TRAINED checkpoint rather than of an initialiser (the released
    `kvnet_scannet.tar` cannot be fetched offline): BatchNorm gamma ~ +-U[0.2, 2.5] (one in ten negative), beta ~ N(0, 0.5);
    every convolution in front of a BatchNorm gets a per-output-channel gain (log-normal) AND a per-output-channel offset
    added to all of its taps, so that behind the non-negative ReLU activations of the previous layer its output has a mean
    far from zero — |mean| / std of the pre-BatchNorm maps reaches ~10 and beyond (oracle/gen_golden.py prints the measured
    distribution): the regime in which a variance taken as E[y^2] - mean^2 loses digits and in which the clamped-FMA ReLU's
    bound is exercised.  Running statistics start away from (0, 1).  Convolutions without a BatchNorm behind them (R-Net,
    the 1x1 head, the K-Net's last layer) keep the initialiser's distribution: their output scale is not normalised away.
    Name-keyed like seeded_state_dict, so the reference model and this package's model receive identical tensors."""
    base = seeded_state_dict(model, seed)
    state = model.state_dict()
    canon = {}
    for name, ref in state.items():
        key = (ref.data_ptr(), tuple(ref.shape))
        canon[key] = min(canon.get(key, name), name)
    dims = {name: ref.dim() for name, ref in state.items()}
    out = {}
    for name, ref in state.items():
        cname = canon[(ref.data_ptr(), tuple(ref.shape))]
        rng = np.random.RandomState((zlib.crc32(("trained/" + cname).encode()) + 7919 * seed) & 0x7FFFFFFF)
        shape = tuple(ref.shape)
        t = base[name]
        stem, _, leaf = name.rpartition(".")
        if leaf == "running_mean":
            t = torch.from_numpy(rng.normal(0, 0.5, shape).astype(np.float32))
        elif leaf == "running_var":
            t = torch.from_numpy(rng.uniform(0.3, 3.0, shape).astype(np.float32))
        elif leaf == "num_batches_tracked":
            t = torch.full((), 1000, dtype=ref.dtype)
        elif leaf == "weight" and ref.dim() >= 4 and stem.endswith(".0") and dims.get(stem[:-2] + ".1.weight") == 1:
            # the convolution of a Sequential(conv, BatchNorm) (children '0', '1')
            col = (shape[0],) + (1,) * (len(shape) - 1)
            gain = np.exp(rng.normal(0, 0.5, col))
            he = float(np.sqrt(2.0 / (int(np.prod(shape[2:])) * shape[0])))
            off = offset * rng.normal(0, 1.0, col) * he
            t = torch.from_numpy((t.numpy().astype(np.float64) * gain + off).astype(np.float32))
        elif name.endswith("lastconv.2.weight") or name == "kv_net.classify.2.weight":
            # the 1x1 head has no BatchNorm behind it: with gammas up to 2.5 in front of it the features (and with them the cost
            # volume, quadratic in the features) would be ~30x the initialiser family's; a trained head keeps the cost in the range
            # sigma_soft_max was tuned for.  0.25 brings BV_cur back to log-probabilities of -100 and above.  The K-Net's last layer
            # (64 -> 1, no BatchNorm either) likewise: its output is a log-probability increment, which a trained net keeps moderate
            t = t * 0.25
        elif leaf == "weight" and ref.dim() == 1:                       # BatchNorm gamma
            sign = np.where(rng.rand(*shape) < 0.1, -1.0, 1.0)
            t = torch.from_numpy((sign * rng.uniform(0.2, 2.5, shape)).astype(np.float32))
        elif leaf == "bias" and dims.get(stem + ".weight") == 1:        # BatchNorm beta
            t = torch.from_numpy(rng.normal(0, 0.5, shape).astype(np.float32))
        out[name] = t
    return out
