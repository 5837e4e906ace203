#!/usr/bin/env python
"""conv2d_wgrad.hip against the vendor weight gradient at the layer shapes of the training iteration (HIP events)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neuralrgbd_amd import ops
dev = "cuda:0"
SHAPES = [(5, 128, 192, 32, 32, 1), (5, 64, 96, 64, 64, 1), (5, 64, 96, 128, 128, 1), (5, 64, 96, 128, 128, 2), (5, 64, 96, 320, 128, 1),
          (5, 64, 96, 64, 128, 1), (1, 64, 96, 128, 128, 1), (1, 128, 192, 96, 96, 1), (1, 256, 384, 64, 64, 1)]


def timed(fn, n=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for N, H, W, ci, co, d in SHAPES:
    x = torch.randn(N, H, W, ci, device=dev)
    gy = torch.randn(N, H, W, co, device=dev)
    ours = timed(lambda: ops.conv2d_wgrad(x, gy, d))
    xn, gn = x.permute(0, 3, 1, 2), gy.permute(0, 3, 1, 2)      # channels-last NCHW views
    vend = timed(lambda: torch.ops.aten.convolution_backward(gn, xn, torch.empty(co, ci, 3, 3, device=dev), None, (1, 1), (d, d), (d, d),
                                                             False, (0, 0), 1, (False, True, False)))
    gf = 2.0 * N * H * W * ci * co * 9 / 1e9
    print("N=%d %3dx%3d %3d->%3d d=%d: ours %7.1f us (%5.1f TF)   vendor %7.1f us (%5.1f TF)" % (N, H, W, ci, co, d, ours, gf / ours * 1e3 / 1e3,
                                                                                             vend, gf / vend * 1e3 / 1e3))
