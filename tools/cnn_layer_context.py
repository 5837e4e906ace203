#!/usr/bin/env python
"""One feature-CNN layer (5 x 192 x 256, 64 -> 64, wino_pc) under different histories: the same launch repeated, with a BatchNorm
finaliser between launches, alternating the plain / residual+materialise instantiations, and as a chain of 16 different layers
(own weights, each reading the previous output) — what a layer costs inside the trunk vs alone.  HIP events, product library."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neuralrgbd_amd import ops

N, H, W, C = 5, 192, 256, 64
g = torch.Generator().manual_seed(0)
dev = "cuda:0"
x = torch.randn(N, H, W, C, generator=g).to(dev)
r = torch.randn(N, H, W, C, generator=g).to(dev)
ws = [ops.conv_wino_pack((torch.randn(C, C, 3, 3, generator=g) * 0.05).to(dev)) for _ in range(16)]
ss = torch.rand(C, 2, generator=g).to(dev)
gamma, beta = torch.ones(C, device=dev), torch.zeros(C, device=dev)
cnt = N * H * W


def timed(fn, n, reps=5):
    fn(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / n)
    return best


def same():
    for _ in range(32):
        ops.conv_wino(x, ws[0], C, 1, 1, x_ss=ss, x_relu=True)

def with_fin():
    for _ in range(32):
        y, st, _ = ops.conv_wino(x, ws[0], C, 1, 1, x_ss=ss, x_relu=True)
        ops.bn_finalize_cm(st, cnt, gamma, beta, 1e-5, 0.1)

def fin_only():
    y, st, _ = ops.conv_wino(x, ws[0], C, 1, 1, x_ss=ss, x_relu=True)
    for _ in range(32):
        ops.bn_finalize_cm(st, cnt, gamma, beta, 1e-5, 0.1)

def alternate():
    for i in range(16):
        ops.conv_wino(x, ws[0], C, 1, 1, x_ss=ss, x_relu=True)
        ops.conv_wino(x, ws[1], C, 1, 1, x_ss=ss, x_relu=False, res=r, res_ss=ss, res_relu=True, materialize=True)

def res_only():
    for i in range(32):
        ops.conv_wino(x, ws[1], C, 1, 1, x_ss=ss, x_relu=False, res=r, res_ss=ss, res_relu=True, materialize=True)

def chain():
    cur, s = x, ss
    for i in range(16):
        cur, st, _ = ops.conv_wino(cur, ws[i], C, 1, 1, x_ss=s, x_relu=True)
        s = ops.bn_finalize_cm(st, cnt, gamma, beta, 1e-5, 0.1)
    for i in range(16):
        cur, st, _ = ops.conv_wino(cur, ws[i], C, 1, 1, x_ss=s, x_relu=True)
        s = ops.bn_finalize_cm(st, cnt, gamma, beta, 1e-5, 0.1)

for name, fn in (("same launch repeated", same), ("+ bn_finalize_cm between", with_fin), ("bn_finalize_cm alone", fin_only),
                 ("residual+materialise repeated", res_only), ("plain / residual alternating", alternate),
                 ("chain of 16 layers (+ finalisers), twice", chain)):
    print("%-44s %7.1f us per step" % (name, timed(fn, 32)))
gr = torch.cuda.CUDAGraph()
s_ = torch.cuda.Stream()
with torch.cuda.stream(s_):
    chain(); torch.cuda.synchronize()
    with torch.cuda.graph(gr, stream=s_):
        chain()
print("%-44s %7.1f us per step" % ("the chain as one hipGraph", timed(gr.replay, 32)))
