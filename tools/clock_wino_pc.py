#!/usr/bin/env python
"""In-kernel clocks of wino_pc.hip (developer build, NRGBD_WINO_ABL=64) for one feature-CNN layer shape (kd = 1): per stage and per
tile, median over workgroups: consumer MFMA time, consumer barrier wait, epilogue; producer publish / transform / barrier."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["NRGBD_WINO_ABL"] = str(64 | int(os.environ.get("EXTRA_ABL", "0")))
from neuralrgbd_amd import _lib
_lib.LIB_PATH = _lib.LIB_PATH.replace("libnrgbd_hip.so", "libnrgbd_hip_dev.so")
from neuralrgbd_amd import ops
N, H, W, Cin, Cout = [int(v) for v in (sys.argv[1:6] if len(sys.argv) > 5 else (5, 192, 256, 64, 64))]
g = torch.Generator().manual_seed(0)
x = torch.randn(N, H, W, Cin, generator=g).cuda()
w = (torch.randn(Cout, Cin, 3, 3, generator=g) * 0.05).cuda()
ss = torch.rand(Cin, 2, generator=g).cuda()
ww = ops.conv_wino_pack32(w) if Cout == 32 else ops.conv_wino_pack(w)
for _ in range(3):
    st = ops.conv_wino(x, ww, Cout, 1, 1, x_ss=ss, x_relu=True)[1]
torch.cuda.synchronize()
ns = Cin // 16
rec = st.reshape(-1)[:256 * 8].reshape(256, 8).double().cpu()
tiles = rec[:, 3]
print("tiles per workgroup: min %d max %d" % (tiles.min().item(), tiles.max().item()))
per_tile = rec[:, [0, 1, 2, 4, 5, 6]] / tiles[:, None]
names = ("mfma", "c-barrier", "epilogue", "publish", "transform", "p-barrier")
print("10 ns ticks per TILE (%d stages), median over workgroups: " % ns + "  ".join("%s %.0f" % (n, v) for n, v in zip(names, per_tile.median(0).values.tolist())))
ab = st.reshape(-1)[8192:8192 + 256 * 4].reshape(256, 4).double().cpu()
t0 = ab[:, 0].min()
rel = (ab[:, :3] - t0) % float(1 << 24)
print("absolute 10 ns ticks from the first workgroup's entry: entry min/med/max %d/%d/%d  loop start %d/%d/%d  loop end %d/%d/%d"
      % tuple(v for c in range(3) for v in (rel[:, c].min().item(), rel[:, c].median().item(), rel[:, c].max().item())))
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    ops.conv_wino(x, ww, Cout, 1, 1, x_ss=ss, x_relu=True)
e1.record(); torch.cuda.synchronize()
print("launch-to-launch %.1f us (developer build)" % (e0.elapsed_time(e1) * 50))
