#!/usr/bin/env python
"""One K-Net 64->64 layer at a SURVEY grid: direct MFMA kernel vs the Winograd-domain kernel (HIP events)."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
GRIDS = {"S": (64, 64, 96), "B": (64, 192, 256), "K": (64, 64, 192), "H": (128, 120, 160)}


def timeit(fn, iters):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="B")
    ap.add_argument("--iters", type=int, default=5)
    args = ap.parse_args()
    from neuralrgbd_amd import ops
    D, H, W = GRIDS[args.config]
    g = torch.Generator().manual_seed(0)
    x = torch.randn(D, H, W, 64, generator=g).cuda()
    r = torch.randn(D, H, W, 64, generator=g).cuda()
    w = (torch.randn(64, 64, 3, 3, 3, generator=g) * 0.05).cuda()
    ss = torch.rand(64, 2, generator=g).cuda()
    wd, ww = ops.conv3d_pack_weights(w), ops.conv3d_wino_pack(w)
    flops = 2.0 * D * H * W * 64 * 64 * 27
    for name, fn in (("direct plain", lambda: ops.conv3d(x, wd, x_ss=ss, x_relu=True)),
                     ("wino   plain", lambda: ops.conv3d_wino(x, ww, x_ss=ss, x_relu=True)),
                     ("direct res+mat", lambda: ops.conv3d(x, wd, x_ss=ss, res=r, materialize=True)),
                     ("wino   res+mat", lambda: ops.conv3d_wino(x, ww, x_ss=ss, res=r, materialize=True))):
        ms = timeit(fn, args.iters)
        print("%-16s %8.3f ms   %6.1f TFLOP/s nominal (27-tap flops)" % (name, ms, flops / ms / 1e9))
    y1 = ops.conv3d(x, wd, x_ss=ss, x_relu=True)[0]
    y2 = ops.conv3d_wino(x, ww, x_ss=ss, x_relu=True)[0]
    print("max|wino - direct| = %.3e (|y|max %.2f)" % ((y1 - y2).abs().max().item(), y1.abs().max().item()))


if __name__ == "__main__":
    main()
