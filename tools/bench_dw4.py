"""wino_dw4.hip (F(4,3) along depth) against wino_dw.hip (F(2,3)) on one 64 -> 64 K-Net layer, HIP events, same chip.
    python tools/bench_dw4.py [B|S|K|H]"""
import sys

import torch

import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neuralrgbd_amd import _lib
if os.environ.get("NRGBD_EXP_LIB"):
    _lib.LIB_PATH = os.environ["NRGBD_EXP_LIB"]
from neuralrgbd_amd import ops

cfg = sys.argv[1] if len(sys.argv) > 1 else "B"
D, H, W = {"B": (64, 192, 256), "S": (64, 64, 96), "K": (64, 64, 192), "H": (128, 120, 160)}[cfg]
dev = "cuda:0"
g = torch.Generator().manual_seed(0)
x = torch.randn(D, H, W, 64, generator=g).to(dev)
w = (torch.randn(64, 64, 3, 3, 3, generator=g) * 0.05).to(dev)
ss = torch.stack((torch.rand(64, generator=g) + 0.5, torch.randn(64, generator=g) * 0.1), 1).to(dev)
w2, w4 = ops.conv_wino_dw_pack(w), ops.conv_wino_dw4_pack(w)
unit = 2.0 ** -12
w2u, w4u = ops.conv_wino_dw_pack(w / unit), ops.conv_wino_dw4_pack(w / unit)


def t(f, n=20, warm=10):
    for _ in range(warm):
        f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


rows = [("ident  wino_dw ", lambda: ops.conv_wino_dw(x, w2, 64)), ("ident  wino_dw4", lambda: ops.conv_wino_dw4(x, w4, 64)),
        ("clamp  wino_dw ", lambda: ops.conv_wino_dw(x, w2u, 64, x_ss=ss, x_relu=True, x_unit=unit)),
        ("clamp  wino_dw4", lambda: ops.conv_wino_dw4(x, w4u, 64, x_ss=ss, x_relu=True, x_unit=unit)),
        ("plain  wino_dw ", lambda: ops.conv_wino_dw(x, w2, 64, x_ss=ss, x_relu=True)),
        ("plain  wino_dw4", lambda: ops.conv_wino_dw4(x, w4, 64, x_ss=ss, x_relu=True))]
for name, f in rows * 2:
    print("%s config %s %dx%dx%d: %.3f ms" % (name, cfg, D, H, W, t(f)))
