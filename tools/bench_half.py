#!/usr/bin/env python
"""The trunk's 32 -> 32 half-resolution layer (5 x 384 x 512 at config B): wino_pc.hip's HALF form vs conv2d.hip's direct form, plain
and with the residual + materialise prologue.  HIP events, product library; with NRGBD_LIB=dev the in-kernel clocks of the HALF form."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neuralrgbd_amd import ops
N, H, W, C = [int(v) for v in sys.argv[1:5]] if len(sys.argv) > 4 else (5, 384, 512, 32)
g = torch.Generator().manual_seed(0)
x = torch.randn(N, H, W, C, generator=g).cuda(); r = torch.randn(N, H, W, C, generator=g).cuda()
ss = torch.rand(C, 2, generator=g).cuda()
w = (torch.randn(C, C, 3, 3, generator=g) * 0.05).cuda()
ww, wd = ops.conv_wino_pack32(w), ops.conv_pack_weights(w)


def timed(fn, n=30):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n

gf = 2.0 * N * H * W * 9 * C * C / 1e9
for name, fn in (("wino HALF plain", lambda: ops.conv_wino(x, ww, C, 1, 1, x_ss=ss, x_relu=True)),
                 ("direct     plain", lambda: ops.conv2d(x, wd, C, 1, x_ss=ss, x_relu=True)),
                 ("wino HALF res+mat", lambda: ops.conv_wino(x, ww, C, 1, 1, x_ss=ss, x_relu=True, res=r, res_ss=ss, materialize=True)),
                 ("direct     res+mat", lambda: ops.conv2d(x, wd, C, 1, x_ss=ss, x_relu=True, res=r, res_ss=ss, materialize=True))):
    t = timed(fn)
    print("%-20s %7.1f us  (%.0f TFLOP/s direct-equivalent)" % (name, t, gf / t * 1e-3 * 1e3 / 1e3 * 1e3 / 1e3 if False else gf / (t * 1e-6) / 1e3))
