#!/usr/bin/env python
"""Micro-benchmark of the sampling kernels alone (no convolutions) at a SURVEY §8(d) grid.

    python tools/bench_costvol.py [--config B] [--iters 20] [--views 4] [--gen lds|gather]

Prints per-kernel mean time from HIP events on the launch stream, algorithmic GB/s (SURVEY §8d
byte counts) and the fraction of the 8 TB/s HBM peak.  Used under rocprofv3 for the PMC passes.
"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
GRIDS = {"S": (64, 96, 64), "B": (192, 256, 64), "K": (64, 192, 64), "H": (120, 160, 128)}


def timeit(fn, iters):
    """Mean duration of a launch, every launch between its OWN pair of HIP events (round 5: one pair around a back-to-back loop adds
    the previous launch's end-of-kernel write-back to every launch — 0.235 vs 0.193 ms for the fused sampling kernel at config B,
    tools/r5_cv_gap.py)."""
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    ev = []
    for _ in range(iters):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        ev.append((e0, e1))
    torch.cuda.synchronize()
    return sum(a.elapsed_time(b) for a, b in ev) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="B")
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--views", type=int, default=4)
    ap.add_argument("--gen", default=None)
    ap.add_argument("--only", default=None, help="costvol|pack|warpvol|resample|softmax")
    ap.add_argument("--dev", action="store_true", help="load libnrgbd_hip_dev.so (python -m neuralrgbd_amd.build --dev): "
                    "honours NRGBD_ABLATE (quad generation: 1 = everything straight from global memory, 2 = staging only, no math)")
    ap.add_argument("--lib", default=None, help="another library next to libnrgbd_hip.so (an A/B build of neuralrgbd_amd.build.build_variant)")
    ap.add_argument("--dslice", default=None, help="a:b — keep only candidates a..b-1 of the depth set (experiments)")
    args = ap.parse_args()
    gen = args.gen
    if args.dev:
        from neuralrgbd_amd import _lib
        _lib.LIB_PATH = _lib.LIB_PATH.replace("libnrgbd_hip.so", "libnrgbd_hip_dev.so")
    if args.lib:
        from neuralrgbd_amd import _lib
        _lib.LIB_PATH = _lib.LIB_PATH.replace("libnrgbd_hip.so", args.lib)
    from neuralrgbd_amd import camera, ops, synth
    from neuralrgbd_amd import homography as H
    h, w, D = GRIDS[args.config]
    V, C = args.views, 67
    dev = "cuda:0"
    cam = camera.scannet_intrinsics(w, h)
    rng = np.random.RandomState(0)
    feats = torch.from_numpy(rng.standard_normal((V + 1, 64, h, w)).astype(np.float32)).to(dev)
    frames = torch.from_numpy(rng.standard_normal((V + 1, 3, 4 * h, 4 * w)).astype(np.float32)).to(dev)
    poses = torch.from_numpy(synth.random_poses(rng, V)).to(dev)
    d_candi = np.linspace(0.1, 5.0, D)
    if args.dslice:
        lo, hi = (int(t) for t in args.dslice.split(":"))
        d_candi = d_candi[lo:hi]
        D = len(d_candi)
    K, rays = H._cam_dev(cam, torch.device(dev))
    d_dev = H._d_candi_dev(d_candi, torch.device(dev))
    KR, Kt = H.homography_terms(K, poses[:, :3, :3], poses[:, :3, 3])
    cx, cy = cam["intrinsic_M"][0, 2], cam["intrinsic_M"][1, 2]
    tex = ops.pack_nhwc(feats, frames)
    bv = ops.costvol(tex[V], tex[:V], KR, Kt, rays, d_dev, cx, cy, 10.0, C, want_cost=False, want_logp=True)[1]
    hw = h * w
    runs = {
        "pack": (lambda: ops.pack_nhwc(feats, frames), 4 * ((V + 1) * 64 * hw + (V + 1) * 3 * 16 * hw + (V + 1) * 68 * hw)),
        "costvol": (lambda: ops.costvol(tex[V], tex[:V], KR, Kt, rays, d_dev, cx, cy, 10.0, C, want_cost=True,
                                        want_logp=False, generation=gen), 4 * ((V + 1) * C * hw + D * hw)),
        "costvol+logsoftmax": (lambda: ops.costvol(tex[V], tex[:V], KR, Kt, rays, d_dev, cx, cy, 10.0, C,
                                                   want_cost=False, want_logp=True, generation=gen), 4 * ((V + 1) * C * hw + D * hw)),
        "warpvol": (lambda: ops.warp_volume(tex[:V, :, :, 64:], (hw * 68, 1, w * 68, 68), tex[V, :, :, 64:],
                                            (1, w * 68, 68), KR, Kt, rays, d_dev, cx, cy, V, 3, h, w, bv_cur=bv, bv_pred=bv),
                    4 * ((V + 1) * 3 * hw + 2 * D * hw + 16 * D * hw)),
        "resample": (lambda: ops.dpv_resample(bv, ops.pose_inverse(poses[2].contiguous()), rays, d_dev, 0.55, 0.42, 2.55, 2.45, -4.16),
                     8 * D * hw),
        "softmax": (lambda: ops.logsoftmax_d(bv, bv), 12 * D * hw),
    }
    for name, (fn, nbytes) in runs.items():
        if args.only and not name.startswith(args.only):
            continue
        ms = timeit(fn, args.iters)
        print("%-20s %8.1f us   %7.1f GB/s algorithmic   %5.2f %% of 8 TB/s" %
              (name, ms * 1e3, nbytes / ms / 1e6, 100 * nbytes / ms / 1e6 / 8000))


if __name__ == "__main__":
    main()
