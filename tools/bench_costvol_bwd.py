#!/usr/bin/env python
"""Backward of the fused cost volume at the training grid (64x96, D = 64, V = 4, C = 67): HIP-event timing.
--dev loads libnrgbd_hip_dev.so, which honours NRGBD_BWD_ABL (1 = no atomics, 2 = no tap loads; results invalid)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if "--dev" in sys.argv:
    from neuralrgbd_amd import _lib
    _lib.LIB_PATH = _lib.LIB_PATH.replace("libnrgbd_hip.so", "libnrgbd_hip_dev.so")
from neuralrgbd_amd import camera, ops, synth
from neuralrgbd_amd import homography as H
h, w, D, V, C = 64, 96, 64, 4, 67
dev = "cuda:0"
cam = camera.scannet_intrinsics(w, h)
rng = np.random.RandomState(0)
feats = torch.from_numpy(rng.standard_normal((V + 1, 64, h, w)).astype(np.float32)).to(dev)
frames = torch.from_numpy(rng.standard_normal((V + 1, 3, 4 * h, 4 * w)).astype(np.float32)).to(dev)
poses = torch.from_numpy(synth.random_poses(rng, V)).to(dev)
K, rays = H._cam_dev(cam, torch.device(dev))
d_dev = H._d_candi_dev(np.linspace(0.1, 5.0, D), torch.device(dev))
KR, Kt = H.homography_terms(K, poses[:, :3, :3], poses[:, :3, 3])
cx, cy = cam["intrinsic_M"][0, 2], cam["intrinsic_M"][1, 2]
tex = ops.pack_nhwc(feats, frames)
g = torch.from_numpy(rng.standard_normal((D, h, w)).astype(np.float32)).to(dev)
fn = lambda: ops.costvol_bwd(tex[V], tex[:V], KR, Kt, rays, d_dev, cx, cy, 10.0, C, g)
fn(); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    fn()
e1.record(); torch.cuda.synchronize()
print("costvol_bwd abl=%s: %.3f ms" % (os.environ.get("NRGBD_BWD_ABL", "0"), e0.elapsed_time(e1) / 10))
