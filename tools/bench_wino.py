#!/usr/bin/env python
"""One K-Net 64->64 layer at a SURVEY grid: direct MFMA kernel vs the Winograd-domain kernel (HIP events)."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
GRIDS = {"S": (64, 64, 96), "B": (64, 192, 256), "K": (64, 64, 192), "H": (128, 120, 160)}


def timeit(fn, iters):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="B")
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--only", default="", help="substring filter on the variant names (e.g. 'wino-dw')")
    ap.add_argument("--cnn", action="store_true", help="also time the feature CNN's 2-D layer shapes")
    ap.add_argument("--dev", action="store_true", help="load libnrgbd_hip_dev.so (python -m neuralrgbd_amd.build --dev): "
                    "NRGBD_WINO_ABL=1|2 (producers only / consumers only) is honoured there")
    args = ap.parse_args()
    if args.dev:
        from neuralrgbd_amd import _lib
        _lib.LIB_PATH = os.environ.get("NRGBD_DEV_LIB") or _lib.LIB_PATH.replace("libnrgbd_hip.so", "libnrgbd_hip_dev.so")   # NRGBD_DEV_LIB: a one-off experimental build
    from neuralrgbd_amd import ops
    D, H, W = GRIDS[args.config]
    g = torch.Generator().manual_seed(0)
    x = torch.randn(D, H, W, 64, generator=g).cuda()
    r = torch.randn(D, H, W, 64, generator=g).cuda()
    w = (torch.randn(64, 64, 3, 3, 3, generator=g) * 0.05).cuda()
    ss = torch.rand(64, 2, generator=g).cuda()
    wd, ww = ops.conv3d_pack_weights(w), ops.conv_wino_pack(w)
    wdw = ops.conv_wino_dw_pack(w)
    flops = 2.0 * D * H * W * 64 * 64 * 27
    for name, fn in (("direct plain", lambda: ops.conv3d(x, wd, x_ss=ss, x_relu=True)),
                     ("wino-pc plain", lambda: ops.conv_wino(x, ww, 64, 3, x_ss=ss, x_relu=True)),
                     ("wino-dw plain", lambda: ops.conv_wino_dw(x, wdw, 64, x_ss=ss, x_relu=True)),
                     ("wino-dw ident", lambda: ops.conv_wino_dw(x, wdw, 64)),
                     ("wino-dw res+mat", lambda: ops.conv_wino_dw(x, wdw, 64, x_ss=ss, res=r, materialize=True)),
                     ("direct res+mat", lambda: ops.conv3d(x, wd, x_ss=ss, res=r, materialize=True)),
                     ("wino-pc res+mat", lambda: ops.conv_wino(x, ww, 64, 3, x_ss=ss, res=r, materialize=True))):
        if args.only and args.only not in name:
            continue
        ms = timeit(fn, args.iters)
        print("%-16s %8.3f ms   %6.1f TFLOP/s nominal (27-tap flops)" % (name, ms, flops / ms / 1e9))
    if args.dev and (int(os.environ.get("NRGBD_WINO_ABL", "0")) & 64):
        # in-kernel clocks (dev build): per workgroup [mfma, consumer barrier, epilogue, tiles | publish, transform, producer barrier]
        st = ops.conv_wino(x, ww, 64, 3, x_ss=ss, x_relu=True)[1]
        torch.cuda.synchronize()
        rec = st.reshape(-1)[:256 * 8].reshape(256, 8).double().cpu()
        per = rec[:, 3:4] * 12
        names = ("mfma", "c-barrier", "epilogue/tile*12", "tiles", "publish", "transform", "p-barrier")
        vals = rec.clone(); vals[:, [0, 1, 2, 4, 5, 6]] /= per
        r2 = st.reshape(-1)[4096:4096 + 256 * 4].reshape(256, 4).double().cpu() / per
        print("publish split: wait-vmcnt %.0f  valu+ds_write %.0f  issue-loads %.0f" % tuple(r2.median(0).values.tolist()[:3]))
        print("in-kernel wall_clock64 (10 ns ticks) per stage, median over workgroups: " + "  ".join("%s %.0f" % (n, v) for n, v in zip(names, vals.median(0).values.tolist())))
    if args.only:
        return
    y1 = ops.conv3d(x, wd, x_ss=ss, x_relu=True)[0]
    y3 = ops.conv_wino(x, ww, 64, 3, x_ss=ss, x_relu=True)[0]
    print("max|wino-pc - direct| = %.3e" % (y1 - y3).abs().max().item())
    if args.cnn:
        import torch.nn.functional as F
        # the feature CNN's layer shapes at config B (5 images): direct conv2d.hip vs the Winograd kernel (kd = 1)
        for (N, Hh, Ww, Cin, Cout, dil) in ((5, 192, 256, 64, 64, 1), (5, 192, 256, 128, 128, 1), (5, 192, 256, 128, 128, 2),
                                            (5, 192, 256, 320, 128, 1), (5, 64, 96, 64, 64, 1), (5, 64, 96, 128, 128, 2)):
            x2 = torch.randn(N, Hh, Ww, Cin, generator=g).cuda()
            w2 = (torch.randn(Cout, Cin, 3, 3, generator=g) * 0.05).cuda()
            s2 = torch.rand(Cin, 2, generator=g).cuda()
            wd2, ww2 = ops.conv_pack_weights(w2), ops.conv_wino_pack(w2)
            fl = 2.0 * N * Hh * Ww * Cin * Cout * 9
            t1 = timeit(lambda: ops.conv2d(x2, wd2, Cout, dil, x_ss=s2, x_relu=True), args.iters)
            t2 = timeit(lambda: ops.conv_wino(x2, ww2, Cout, 1, dil, x_ss=s2, x_relu=True), args.iters)
            d = (ops.conv2d(x2, wd2, Cout, dil, x_ss=s2, x_relu=True)[0] - ops.conv_wino(x2, ww2, Cout, 1, dil, x_ss=s2, x_relu=True)[0]).abs().max().item()
            print("conv2d N%d %dx%d %3d->%3d dil%d: direct %7.3f ms (%5.1f TF)  wino-pc %7.3f ms (%5.1f TF nominal)  max|d| %.2e"
                  % (N, Hh, Ww, Cin, Cout, dil, t1, fl / t1 / 1e9, t2, fl / t2 / 1e9, d))


if __name__ == "__main__":
    main()
