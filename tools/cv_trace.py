#!/usr/bin/env python
"""Phase clocks of the fused sampling kernel (costvol_quad), per workgroup, from the DEVELOPER build.

    python -m neuralrgbd_amd.build --dev && python tools/cv_trace.py [--config B]

libnrgbd_hip_dev.so (-DNRGBD_DEV) lets thread 0 of every workgroup accumulate shader clocks per phase and write them to the
buffer whose address is in NRGBD_CV_TRACE (csrc/costvol_quad.hip, CVT_* macros; compiled out of the product library).  Printed:
where a workgroup's time goes (footprint boxes, run selection, LDS-DMA issue, the wait for it, tap math of staged / unstaged runs,
log-softmax), how the candidates split into runs, and the spread of the workgroups' end times (the tail of the launch).
"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
GRIDS = {"S": (64, 96, 64), "B": (192, 256, 64), "K": (64, 192, 64), "H": (120, 160, 128)}
SLOTS = {3: "footprint boxes", 4: "run selection", 16: "barrier before a fill", 5: "LDS-DMA issue", 6: "wait for the fill (vmcnt + barrier)",
         7: "tap math, staged runs (incl. RMW)", 8: "tap math, unstaged groups", 9: "log-softmax"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="B")
    ap.add_argument("--views", type=int, default=4)
    ap.add_argument("--lib", default="libnrgbd_hip_dev.so")
    ap.add_argument("--map", action="store_true", help="print the per-tile cost map (one-chunk launches)")
    ap.add_argument("--seed", type=int, default=0, help="seed of the synthetic poses / features")
    args = ap.parse_args()
    from neuralrgbd_amd import _lib
    _lib.LIB_PATH = _lib.LIB_PATH.replace("libnrgbd_hip.so", args.lib)
    from neuralrgbd_amd import camera, ops, synth
    from neuralrgbd_amd import homography as H
    h, w, D = GRIDS[args.config]
    V, C = args.views, 67
    dev = torch.device("cuda:0")
    cam = camera.scannet_intrinsics(w, h) if args.config != "K" else camera.kitti_intrinsics(w, h)
    rng = np.random.RandomState(args.seed)
    feats = torch.from_numpy(rng.standard_normal((V + 1, 64, h, w)).astype(np.float32)).to(dev)
    frames = torch.from_numpy(rng.standard_normal((V + 1, 3, 4 * h, 4 * w)).astype(np.float32)).to(dev)
    poses = torch.from_numpy(synth.random_poses(rng, V)).to(dev)
    d_candi = np.linspace(0.1, 5.0, D) if args.config != "K" else np.linspace(1.0, 60.0, D)
    K, rays = H._cam_dev(cam, dev)
    d_dev = H._d_candi_dev(d_candi, dev)
    KR, Kt = H.homography_terms(K, poses[:, :3, :3], poses[:, :3, 3])
    cx, cy = cam["intrinsic_M"][0, 2], cam["intrinsic_M"][1, 2]
    tex = ops.pack_nhwc(feats, frames)
    run = lambda: ops.costvol(tex[V], tex[:V], KR, Kt, rays, d_dev, cx, cy, 10.0, C, want_cost=False, want_logp=True)
    for _ in range(30):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(100):
        run()
    e1.record()
    torch.cuda.synchronize()
    print("untraced: %.1f us per launch" % (e0.elapsed_time(e1) * 10))
    trace = torch.zeros((8192, 24), dtype=torch.int64, device=dev)
    os.environ["NRGBD_CV_TRACE"] = str(trace.data_ptr())
    for _ in range(5):
        run()
    torch.cuda.synchronize()
    e0.record()
    for _ in range(20):
        run()
    e1.record()
    torch.cuda.synchronize()
    print("traced:   %.1f us per launch" % (e0.elapsed_time(e1) * 50))
    t = trace.cpu().numpy()
    t = t[t[:, 2] > 0]
    n = len(t)
    tot = t[:, 2].astype(np.float64)
    print("%d workgroups; shader clocks per workgroup: min %.0f  mean %.0f  max %.0f" % (n, tot.min(), tot.mean(), tot.max()))
    acc = 0.0
    for slot, name in SLOTS.items():
        v = t[:, slot].astype(np.float64)
        acc += v.mean()
        print("  %-40s mean %8.0f clk = %5.1f %%   (min %.0f max %.0f)" % (name, v.mean(), 100 * v.mean() / tot.mean(), v.min(), v.max()))
    print("  %-40s mean %8.0f clk = %5.1f %%" % ("unaccounted (prologue, loop glue)", tot.mean() - acc, 100 * (tot.mean() - acc) / tot.mean()))
    runs = t[:, 10:14].astype(np.float64).mean(0)
    print("per workgroup (all views): staged runs of 16 / 8 / 4 / 2 candidates: %.1f / %.1f / %.1f / %.1f; unstaged groups %.1f; staged candidates %.1f of %d; patch texels filled %.0f"
          % (runs[0], runs[1], runs[2], runs[3], t[:, 14].mean(), t[:, 15].mean(), D * V, t[:, 19].mean()))
    # the launch's timeline on the 100 MHz wall clock
    s0 = t[:, 0].min()
    start, end = (t[:, 0] - s0) / 100.0, (t[:, 1] - s0) / 100.0          # us
    print("wall clock: workgroup starts %.1f .. %.1f us, ends %.1f .. %.1f us (mean %.1f, median %.1f, p10 %.1f, p90 %.1f)" %
          (start.min(), start.max(), end.min(), end.max(), end.mean(), np.median(end), np.percentile(end, 10), np.percentile(end, 90)))
    dur = end - start
    print("workgroup duration: min %.1f  mean %.1f  max %.1f us; launch span %.1f us => mean residency %.2f of the span" %
          (dur.min(), dur.mean(), dur.max(), end.max(), dur.mean() / end.max()))
    # per compute unit: (xcc, se, cu) from HW_ID
    hw = t[:, 17]
    cu = ((t[:, 18] & 0xF) << 16) | (((hw >> 13) & 0x7) << 8) | ((hw >> 8) & 0xF) | (((hw >> 12) & 0x1) << 12)
    ids = np.unique(cu)
    last = np.array([end[cu == c].max() for c in ids])
    cnt = np.array([(cu == c).sum() for c in ids])
    print("%d distinct (xcc, se, sh, cu) ids; workgroups per id min %d max %d; last end per id: min %.1f mean %.1f max %.1f us" %
          (len(ids), cnt.min(), cnt.max(), last.min(), last.mean(), last.max()))
    xcc = t[:, 18] & 0xF
    print("per XCD: last end (us) " + " ".join("%d:%.0f" % (x, end[xcc == x].max()) for x in np.unique(xcc)) +
          "   mean workgroup clocks " + " ".join("%d:%.0fk" % (x, tot[xcc == x].mean() / 1e3) for x in np.unique(xcc)))
    if args.map and n == ((h + 7) // 8) * ((w + 7) // 8) and n % 8 == 0:
        # cost map: workgroup b -> tile as costvol_quad.hip maps it (one chunk per tile, XCD bands, scrambled inside the band)
        per, tiles_x = n // 8, (w + 7) // 8
        grid = np.zeros(n)
        for b in range(n):
            u = b >> 3
            if per % 32 == 0:
                r, c = u >> 5, u & 31
                u = (r << 5) | ((c * 5 + r * 11) & 31)
            grid[(b & 7) * per + u] = tot[b]
        print("cost map (k clocks per tile, rows = tile rows):")
        for row in grid.reshape(-1, tiles_x):
            print(" ".join("%3d" % round(v / 1e3) for v in row))
    # cost by tile position (rows of the image): how uneven is the work
    order = np.argsort(tot)
    print("slowest 5 workgroups (block id, clocks, unstaged groups, runs16/8/4/2):", [(int(i), int(tot[i]), int(t[i, 14]), tuple(int(x) for x in t[i, 10:14])) for i in order[-5:]])
    print("fastest 5 workgroups:", [(int(i), int(tot[i]), int(t[i, 14]), tuple(int(x) for x in t[i, 10:14])) for i in order[:5]])


if __name__ == "__main__":
    main()
