#!/usr/bin/env python
"""The K-Net's first layer (16 -> 64, models/basic.py:113-117) at a K-Net grid: wino_dw.hip (Winograd along depth too: 4 stages and 4
folds per tile pair) vs wino_pc.hip kd = 3 (3 stages per tile, no fold) vs the direct kernel.  HIP events, steady state."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neuralrgbd_amd import ops
D, H, W = [int(v) for v in sys.argv[1:4]] if len(sys.argv) > 3 else (64, 192, 256)
g = torch.Generator().manual_seed(0)
x = torch.randn(D, H, W, 16, generator=g).cuda()
w = (torch.randn(64, 16, 3, 3, 3, generator=g) * 0.05).cuda()
wdw, wpc, wd = ops.conv_wino_dw_pack(w), ops.conv_wino_pack(w), ops.conv3d_pack_weights(w)


def t(fn, n=30):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n

a = ops.conv_wino_dw(x, wdw, 64)[0]
b = ops.conv_wino(x, wpc, 64, 3)[0]
print("max|wino_dw - wino_pc| %.2e (|y|max %.2f)" % ((a - b).abs().max().item(), a.abs().max().item()))
print("first layer %dx%dx%d 16->64: wino_dw %.3f ms | wino_pc kd=3 %.3f ms | direct %.3f ms" %
      (D, H, W, t(lambda: ops.conv_wino_dw(x, wdw, 64)), t(lambda: ops.conv_wino(x, wpc, 64, 3)), t(lambda: ops.conv3d(x, wd))))
