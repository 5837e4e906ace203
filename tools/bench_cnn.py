#!/usr/bin/env python
"""Feature-CNN timing on the GPU: every conv shape of the trunk (matrix-core kernel vs the vendor library) and the
whole trunk (forward_channels_last vs forward), at a §8(d) configuration.  python tools/bench_cnn.py [--config B]"""
import argparse
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import CONFIGS  # noqa: E402
from neuralrgbd_amd import nets, ops  # noqa: E402


def timeit(fn, n=10):
    fn(); fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="B")
    args = ap.parse_args()
    cfg = CONFIGS[args.config]
    H, W = cfg["H"], cfg["W"]
    dev = torch.device("cuda:0")
    N = 5
    shapes = [("stem/layer1 32->32", H // 2, W // 2, 32, 32, 1, 8), ("layer2 64->64", H // 4, W // 4, 64, 64, 1, 31),
              ("layer3 64->128", H // 4, W // 4, 64, 128, 1, 1), ("layer3 128->128", H // 4, W // 4, 128, 128, 1, 5),
              ("layer4 128->128 d2", H // 4, W // 4, 128, 128, 2, 6), ("lastconv 320->128", H // 4, W // 4, 320, 128, 1, 1)]
    total_m = total_v = 0.0
    for name, h, w, ci, co, d, count in shapes:
        x = torch.randn(N, h, w, ci, device=dev)
        r = torch.randn(N, h, w, ci, device=dev)
        ss = torch.rand(ci, 2, device=dev)
        wt = torch.randn(co, ci, 3, 3, device=dev) * 0.05
        wp = ops.conv_pack_weights(wt)
        xn = x.permute(0, 3, 1, 2).contiguous()
        t_plain = timeit(lambda: ops.conv2d(x, wp, co, d, x_ss=ss, x_relu=True))
        t_res = timeit(lambda: ops.conv2d(x, wp, co, d, x_ss=ss, res=r, materialize=True))
        t_v = timeit(lambda: F.conv2d(xn, wt, padding=d, dilation=d))
        gf = 2.0 * N * h * w * 9 * ci * co / 1e9
        print("%-22s %4dx%-4d  mfma %.3f ms (%.0f TF)  +res/mat %.3f ms (%.0f TF)  vendor %.3f ms (%.0f TF)   x%d"
              % (name, h, w, t_plain, gf / t_plain, t_res, gf / t_res, t_v, gf / t_v, count))
        total_m += count * 0.5 * (t_plain + t_res)
        total_v += count * t_v
    print("sum over the trunk's 3x3 stride-1 convs: mfma %.2f ms, vendor conv only %.2f ms" % (total_m, total_v))
    torch.manual_seed(0)
    fe = nets.FeatureExtractor(feature_dim=64, multi_scale=True).to(dev)
    x = torch.rand(N, 3, H, W, device=dev)
    with torch.no_grad():
        t_new = timeit(lambda: fe.forward_channels_last(x), 5)
        t_old = timeit(lambda: fe(x), 5)
    print("trunk: matrix-core path %.2f ms, vendor path %.2f ms" % (t_new, t_old))


if __name__ == "__main__":
    main()
