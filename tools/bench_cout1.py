"""Timing of the K-Net 64->1 layer (conv3d_cout1: depth-marching tap projection) at a K-Net grid; with the developer library
(NRGBD_LIB=dev) NRGBD_C1_NZ sweeps the number of depth chunks.  python tools/bench_cout1.py [D H W]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neuralrgbd_amd import _lib
if os.environ.get("NRGBD_LIB") == "dev":
    _lib.LIB_PATH = _lib.LIB_PATH.replace("libnrgbd_hip.so", "libnrgbd_hip_dev.so")
from neuralrgbd_amd import ops
D, H, W = [int(v) for v in sys.argv[1:4]] if len(sys.argv) > 3 else (64, 192, 256)
x = torch.randn(D, H, W, 64, device="cuda"); ss = torch.randn(64, 2, device="cuda"); w = torch.randn(27, 64, device="cuda") * 0.05
f = lambda: ops.conv3d_cout1(x, w, x_ss=ss, x_relu=True)
for _ in range(20):
    f()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    f()
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 20
print("cout1 %dx%dx%d NZ=%s: %.3f ms = %.2f TB/s of input" % (D, H, W, os.environ.get("NRGBD_C1_NZ", "auto"), ms, D * H * W * 256 / ms / 1e9))
