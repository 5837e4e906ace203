#!/usr/bin/env python
"""conv3d weight gradient at the training grid (64x64x96, 64 -> 64 and 16 -> 64): HIP events, steady state, vs float64."""
import os, sys, torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neuralrgbd_amd import ops
for (D, H, W, Cin) in ((64, 64, 96, 64), (64, 64, 96, 16), (6, 12, 40, 64)):
    g = torch.Generator().manual_seed(0)
    x = torch.randn(D, H, W, Cin, generator=g).cuda(); gy = torch.randn(D, H, W, 64, generator=g).cuda()
    for _ in range(10): dw = ops.conv3d_wgrad(x, gy)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(30): dw = ops.conv3d_wgrad(x, gy)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 30
    if D * H * W <= 64 * 64 * 96:
        xs = x.permute(3, 0, 1, 2)[None].double().cpu(); gs = gy.permute(3, 0, 1, 2)[None].double().cpu()
        if D > 16:   # float64 reference on a depth slab to keep the host time short
            xs, gs = xs[:, :, :10], gs[:, :, :10]
            dws = ops.conv3d_wgrad(x[:10].contiguous(), gy[:10].contiguous())
        else:
            dws = dw
        w = torch.zeros(64, Cin, 3, 3, 3, dtype=torch.float64, requires_grad=True)
        F.conv3d(xs, w, padding=1).backward(gs)
        err = (dws.double().cpu() - w.grad).abs().max().item() / w.grad.abs().max().item()
    print("conv3d_wgrad %dx%dx%d Cin=%d: %.3f ms  (%.1f TFLOP/s direct-equivalent)  rel. err vs fp64 %.2e" %
          (D, H, W, Cin, ms, 2.0 * D * H * W * Cin * 64 * 27 / ms / 1e9, err))
