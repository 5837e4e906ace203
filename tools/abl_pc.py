#!/usr/bin/env python
"""wino_pc.hip on the developer library: full kernel vs consumers only (NRGBD_WINO_ABL=2: no producer work) vs producers only (1),
for an R-Net full-resolution layer (2 x 768 x 1024, 80 -> 64) and a trunk layer (5 x 192 x 256, 64 -> 64): how much of a layer's time
the producers cost.  python tools/abl_pc.py"""
import os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import torch
    from neuralrgbd_amd import _lib
    _lib.LIB_PATH = _lib.LIB_PATH.replace("libnrgbd_hip.so", "libnrgbd_hip_dev.so")
    from neuralrgbd_amd import ops
    g = torch.Generator().manual_seed(0)

    def t(fn, n=20):
        for _ in range(10):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e3 / n
    x = torch.randn(2, 768, 1024, 80, generator=g).cuda()
    w = ops.conv_wino_pack((torch.randn(64, 80, 3, 3, generator=g) * 0.05).cuda())
    b = torch.zeros(64).cuda()
    out = torch.empty(2, 768, 1024, 64).cuda()
    r = t(lambda: ops.conv_wino_rnet(x, w, 64, bias=b, out=out))
    x2 = torch.randn(5, 192, 256, 64, generator=g).cuda()
    w2 = ops.conv_wino_pack((torch.randn(64, 64, 3, 3, generator=g) * 0.05).cuda())
    ss = torch.rand(64, 2, generator=g).cuda()
    c = t(lambda: ops.conv_wino(x2, w2, 64, 1, 1, x_ss=ss, x_relu=True))
    print("abl=%-3s R-Net 80->64 @2x768x1024: %7.1f us   trunk 64->64 @5x192x256: %6.1f us" % (os.environ.get("NRGBD_WINO_ABL", "0"), r, c))
else:
    for abl in sys.argv[1:] or ["0", "2", "1"]:
        subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=dict(os.environ, NRGBD_WINO_ABL=abl))
