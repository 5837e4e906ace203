"""TEST INFRASTRUCTURE ONLY — numerics of a Gram-matrix formulation of the plane-sweep cost (round 6), on the CPU.

    python -m oracle.costvol_gram_eval [trained]

csrc/costvol_quad.hip evaluates, per (pixel p, candidate k, view v), cost = sum_c (a_c - sum_j w_j t_jc)^2 directly (warping/homography.py
:81-87,:293-331): 4 taps x C channels of FMAs on the vector ALUs — the kernel is VALU-issue bound at 118 flop per algorithmic byte
(profiles/r5_costvol_limits.txt).  The same number expands to

    cost = a.a - 2 sum_j w_j (a.t_j) + sum_j sum_l w_j w_l (t_j.t_l)

where a.t (reference pixel x source texel, a 64 x 156 x 67 matrix product per tile and patch) and t.t (texel x its 2x2 neighbours) are
matrix-core work and the per-(pixel, candidate) part shrinks to 14 gathers + ~20 scalar FMAs.  The catch is cancellation: at the matching
candidate cost << a.a, and the three terms carry rounding errors of ~C eps |a||t| each.  This script measures what that does to the cost
and to BV = log_softmax(-sum_v cost_v / sigma) on the config-S windows, every dot product accumulated sequentially in float32 (what an
fp32 matrix core does along K at best).  Result: profiles/r6_costvol_limits.txt.
"""
import sys

import numpy as np
import torch

import neuralrgbd_amd
from neuralrgbd_amd import camera, synth
from oracle import cpu_oracle as co, kvnet_oracle as ko


def seqdot(x, y):
    """sum over axis 0 of x * y, accumulated sequentially in float32."""
    acc = np.zeros(np.broadcast(x[0], y[0]).shape, np.float32)
    for c in range(x.shape[0]):
        acc = (acc + x[c] * y[c]).astype(np.float32)
    return acc


def coords(KR, Kt, rays, d, cx, cy, w, h):
    """sweep_coords + unnormalize (align_corners False) of oracle/nrgbd_oracle.c in float32 numpy (FMA contraction aside)."""
    f = np.float32
    rx, ry, rz = rays[0], rays[1], rays[2]
    t2x = KR[0] * rx + KR[1] * ry + KR[2] * rz
    t2y = KR[3] * rx + KR[4] * ry + KR[5] * rz
    t2z = KR[6] * rx + KR[7] * ry + KR[8] * rz
    px, py, pz = Kt[0] + t2x * f(d), Kt[1] + t2y * f(d), Kt[2] + t2z * f(d)
    den = pz + f(1e-10)
    gx, gy = (px / den - f(cx)) / f(cx), (py / den - f(cy)) / f(cy)
    return ((gx + f(1)) * f(w) - f(1)) / f(2), ((gy + f(1)) * f(h) - f(1)) / f(2)


def main():
    trained = len(sys.argv) > 1
    H, W, D = 256, 384, 64
    cam = camera.scannet_intrinsics(W // 4, H // 4)
    d_candi = np.linspace(0.1, 5.0, D)
    sigma = 10.0
    model = neuralrgbd_amd.KVNET(64, cam, d_candi, sigma, 64, None, if_refined=True, refineNet_name="DPV", t_win_r=2)
    sd = synth.trained_like_state_dict(model, 0) if trained else synth.seeded_state_dict(model, 0)
    ref, src, poses = synth.noise_window(102, H, W)
    torch.set_num_threads(8)
    with torch.no_grad():
        BV, feats, full = ko.dnet(sd, ref, src, poses, cam, d_candi, sigma)
    full = full.numpy().astype(np.float32)                    # [V+1, 67, h, w]
    V = src.shape[1]
    C, h, w = full.shape[1:]
    a = full[V].reshape(C, h * w)
    KR, Kt = ko._terms(cam, poses[0])
    rays = cam["unit_ray_array_2D"].numpy().astype(np.float32)
    cx, cy = float(cam["intrinsic_M"][0, 2]), float(cam["intrinsic_M"][1, 2])
    print("features: |a| rms %.3f  max %.3f   a.a mean %.2f max %.2f" % (np.sqrt((a ** 2).mean()), np.abs(a).max(), (a * a).sum(0).mean(),
                                                                     (a * a).sum(0).max()))
    aa = seqdot(a, a)
    cost_d = np.zeros((D, h * w), np.float32)
    cost_g = np.zeros((D, h * w), np.float32)
    cost_64 = np.zeros((D, h * w), np.float64)
    for v in range(V):
        t = np.concatenate((full[v].reshape(C, h * w), np.zeros((C, 1), np.float32)), 1)      # texel h*w = the zero texel
        for k in range(D):
            ix, iy = coords(KR[v], Kt[v], rays, d_candi[k], cx, cy, w, h)
            x0, y0 = np.floor(ix), np.floor(iy)
            fx, fy = ix - x0, iy - y0
            ws, idx = [], []
            for dy, wy in ((0, np.float32(1) - fy), (1, fy)):
                for dx, wx in ((0, np.float32(1) - fx), (1, fx)):
                    xs, ys = x0 + dx, y0 + dy
                    ok = (xs >= 0) & (xs <= w - 1) & (ys >= 0) & (ys <= h - 1)
                    idx.append(np.where(ok, ys * w + xs, h * w).astype(np.int64))
                    ws.append((wy * wx).astype(np.float32))
            taps = [t[:, i] for i in idx]                                                    # 4 x [C, hw]
            # direct form, float32, the oracle's order
            s = ((taps[0] * ws[0] + taps[1] * ws[1]) + taps[2] * ws[2]) + taps[3] * ws[3]
            df = s - a
            acc_d = seqdot(df, df)
            s64 = sum(tp.astype(np.float64) * wt.astype(np.float64) for tp, wt in zip(taps, ws))
            acc_64 = ((s64 - a) ** 2).sum(0)
            # Gram form, every dot product sequential in float32
            g = [seqdot(a, tp) for tp in taps]
            cross = np.zeros(h * w, np.float32)
            for j in range(4):
                cross = (cross + ws[j] * g[j]).astype(np.float32)
            quad = np.zeros(h * w, np.float32)
            for j in range(4):
                for l in range(j, 4):
                    tt = seqdot(taps[j], taps[l])
                    quad = (quad + (np.float32(1 if j == l else 2) * ws[j] * ws[l]) * tt).astype(np.float32)
            acc_g = ((aa - np.float32(2) * cross) + quad).astype(np.float32)
            cost_d[k] = cost_d[k] + acc_d / np.float32(sigma)
            cost_g[k] = cost_g[k] + acc_g / np.float32(sigma)
            cost_64[k] += acc_64 / sigma
        print("view %d done" % v, flush=True)
    ls = lambda c: torch.log_softmax(-torch.from_numpy(np.asarray(c)), 0).numpy()
    bd, bg, b64 = ls(cost_d), ls(cost_g), ls(cost_64)
    c_or = co.costvol(full[V], full[:V], KR, Kt, rays, d_candi, cx, cy, sigma).reshape(D, h * w)
    print("direct numpy vs C oracle cost: max %.3e" % np.abs(cost_d - c_or).max())
    print("cost (sum over %d views / sigma): mean %.3f  min %.4f" % (V, cost_64.mean(), cost_64.min()))
    for name, c, b in (("direct fp32", cost_d, bd), ("Gram fp32  ", cost_g, bg)):
        ec, eb = np.abs(c - cost_64), np.abs(b - b64)
        print("%s  cost vs fp64: max %.3e mean %.3e | BV vs fp64: max %.3e mean %.3e | arg-max flips %d of %d" %
              (name, ec.max(), ec.mean(), eb.max(), eb.mean(), int((b.argmax(0) != b64.argmax(0)).sum()), h * w))
    print("Gram vs direct BV: max %.3e  L1 %.3e" % (np.abs(bg - bd).max(), np.abs(bg - bd).mean()))


if __name__ == "__main__":
    main()
