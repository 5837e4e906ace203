import os, sys, torch
sys.path.insert(0, os.getcwd())
from neuralrgbd_amd import _lib
if len(sys.argv) > 1: _lib.LIB_PATH = _lib.LIB_PATH.replace("libnrgbd_hip.so", sys.argv[1])
from neuralrgbd_amd import ops
for (D, H, W, Cin) in ((64, 64, 96, 64), (64, 64, 96, 16)):
    g = torch.Generator().manual_seed(0)
    x = torch.randn(D, H, W, Cin, generator=g).cuda(); gy = torch.randn(D, H, W, 64, generator=g).cuda()
    for _ in range(10): dw = ops.conv3d_wgrad(x, gy)
    ev = []
    for _ in range(30):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); dw = ops.conv3d_wgrad(x, gy); e1.record(); ev.append((e0, e1))
    torch.cuda.synchronize()
    print(os.path.basename(_lib.LIB_PATH), "conv3d_wgrad Cin=%d: %.3f ms  checksum %.6f" % (Cin, sum(a.elapsed_time(b) for a, b in ev) / 30, dw.double().abs().mean().item()))
