#!/usr/bin/env python
"""wino_pc.hip's HALF form (32 -> 32 @ 5 x 384 x 512) on the developer library: full kernel, consumers only (NRGBD_WINO_ABL=2), producers
only (1), and the consumer's in-kernel clocks (64: MFMA loop / barrier wait / epilogue shares).  python tools/abl_half.py"""
import os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import torch
    from neuralrgbd_amd import _lib
    _lib.LIB_PATH = _lib.LIB_PATH.replace("libnrgbd_hip.so", "libnrgbd_hip_dev.so")
    from neuralrgbd_amd import ops
    g = torch.Generator().manual_seed(0)
    x = torch.randn(5, 384, 512, 32, generator=g).cuda()
    ss = torch.rand(32, 2, generator=g).cuda()
    ww = ops.conv_wino_pack32((torch.randn(32, 32, 3, 3, generator=g) * 0.05).cuda())
    fn = lambda: ops.conv_wino(x, ww, 32, 1, 1, x_ss=ss, x_relu=True)
    for _ in range(10): y, st, _ = fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): y, st, _ = fn()
    e1.record(); torch.cuda.synchronize()
    abl = int(os.environ.get("NRGBD_WINO_ABL", "0"))
    msg = "abl=%-3d HALF 32->32 @5x384x512: %7.1f us" % (abl, e0.elapsed_time(e1) * 1e3 / 20)
    if abl & 64:
        o = st.reshape(-1)[:8].tolist()
        tot = o[0] + o[1] + o[2]
        msg += "   consumer clocks of workgroup 0: mfma loop %.0f (%.0f %%), barrier %.0f (%.0f %%), epilogue %.0f (%.0f %%), %d tiles" % (
            o[0], 100 * o[0] / tot, o[1], 100 * o[1] / tot, o[2], 100 * o[2] / tot, int(o[3]))
    print(msg)
else:
    for abl in sys.argv[1:] or ["0", "64", "2", "1"]:
        subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=dict(os.environ, NRGBD_WINO_ABL=abl))
