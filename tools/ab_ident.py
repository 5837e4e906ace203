import os, sys, torch
sys.path.insert(0, "/root/repo")
from neuralrgbd_amd import ops
D, H, W = 64, 192, 256
g = torch.Generator().manual_seed(0)
x = torch.randn(D, H, W, 64, generator=g).cuda()
w = (torch.randn(64, 64, 3, 3, 3, generator=g) * 0.05).cuda()
ss = torch.rand(64, 2, generator=g).cuda()
wdw = ops.conv_wino_dw_pack(w)
def t(fn, n=40):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
plain = lambda: ops.conv_wino_dw(x, wdw, 64, x_ss=ss, x_relu=True)
ident = lambda: ops.conv_wino_dw(x, wdw, 64)
for _ in range(40): plain()
for rnd in range(3):
    print("plain %.3f  ident %.3f  plain %.3f  ident %.3f" % (t(plain), t(ident), t(plain), t(ident)))
