#!/usr/bin/env python
"""Ablation timings of wino_dw.hip on the dev library (python -m neuralrgbd_amd.build --dev): what bounds a stage.
NRGBD_WINO_ABL bits: 1 no MFMAs, 2 no producer work, 4 no transform, 8 no publish A, 16 no publish B, 32 no refills, 64 no fold."""
import os
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import torch
    from neuralrgbd_amd import _lib
    _lib.LIB_PATH = _lib.LIB_PATH.replace("libnrgbd_hip.so", "libnrgbd_hip_dev.so")
    from neuralrgbd_amd import ops
    D, H, W = 64, 192, 256
    g = torch.Generator().manual_seed(0)
    x = torch.randn(D, H, W, 64, generator=g).cuda()
    r = torch.randn(D, H, W, 64, generator=g).cuda()
    w = (torch.randn(64, 64, 3, 3, 3, generator=g) * 0.05).cuda()
    ss = torch.rand(64, 2, generator=g).cuda()
    wdw = ops.conv_wino_dw_pack(w)

    def t(fn, it=5):
        fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(it):
            fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / it
    a = t(lambda: ops.conv_wino_dw(x, wdw, 64, x_ss=ss, x_relu=True))
    b = t(lambda: ops.conv_wino_dw(x, wdw, 64, x_ss=ss, res=r, materialize=True))
    print("abl=%-4s plain %.3f ms  res+mat %.3f ms" % (os.environ.get("NRGBD_WINO_ABL", "0"), a, b))
else:
    for abl in sys.argv[1:] or ["0", "1", "2", "4", "8", "16", "24", "32", "64", "3"]:
        env = dict(os.environ, NRGBD_WINO_ABL=abl)
        subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=env)
