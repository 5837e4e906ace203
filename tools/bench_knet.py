#!/usr/bin/env python
"""Micro-benchmark of the K-Net kernels (conv3d.hip) at a SURVEY §8(d) grid: per-layer time and TFLOP/s."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
GRIDS = {"S": (64, 64, 96), "B": (64, 192, 256), "K": (64, 64, 192), "H": (128, 120, 160)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="B")
    ap.add_argument("--iters", type=int, default=5)
    args = ap.parse_args()
    from neuralrgbd_amd import ops
    D, H, W = GRIDS[args.config]
    dev = "cuda:0"
    x = torch.randn(D, H, W, 64, device=dev)
    res = torch.randn(D, H, W, 64, device=dev)
    w = torch.randn(64, 64, 3, 3, 3, device=dev) * 0.05
    ss = torch.randn(64, 2, device=dev)
    wp = ops.conv3d_pack_weights(w)
    y = torch.empty_like(x)
    mat = torch.empty_like(x)
    cases = {
        "conv3d 64->64 plain": lambda: ops.conv3d(x, wp, out=y),
        "conv3d 64->64 bn+relu prologue": lambda: ops.conv3d(x, wp, x_ss=ss, x_relu=True, out=y),
        "conv3d 64->64 bn + residual + materialize": lambda: ops.conv3d(x, wp, x_ss=ss, res=res, materialize=True, out=y, mat_out=mat),
    }
    flops = 2.0 * D * H * W * 64 * 64 * 27
    for name, fn in cases.items():
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(args.iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / args.iters
        print("%-45s %7.3f ms  %6.1f TFLOP/s  (%.0f %% of the 157.3 TFLOP/s fp32 matrix peak)" %
              (name, ms, flops / ms / 1e9, 100 * flops / ms / 1e9 / 157.3))


if __name__ == "__main__":
    main()
