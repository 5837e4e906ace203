"""TEST INFRASTRUCTURE ONLY — numerics of a cheaper-multiply Winograd form for the K-Net, on the CPU.

    python -m oracle.wino_f4_eval [S]

The K-Net's 3x3x3 layers run on csrc/wino_dw.hip as F(2x2, 3x3) in the plane x F(2, 3) along depth (8 multiplies per output
voxel).  F(4x4, 3x3) in the plane x F(2, 3) along depth would need 4.5 — 1.78x fewer matrix-core flops — but its transforms carry
entries up to 8 and 1/24 and amplify fp32 rounding.  Before any kernel is written this script runs the update frame of the config-S
two-frame windows (the parity test's) with the K-Net's convolutions EMULATED in float32 torch, operation for operation as a kernel
would do them (input transform in fp32, channel contraction in fp32, inverse transform in fp32), three ways:
    direct   F.conv3d (the oracle's own: what the reference computes)
    F2       F(2x2,3x3) x F(2,3)   — today's kernel form (calibrates the emulation against the measured GPU numbers)
    F4       F(4x4,3x3) x F(2,3)   — the candidate
and reports DPV's distance from the oracle (fp32 direct) and from the same graph in float64.
Acceptance (VERDICT): DPV L1 vs the oracle < 5e-5 and |F4 - fp64| <= 1.25 |oracle - fp64|.
"""
import sys

import numpy as np
import torch
import torch.nn.functional as F

from neuralrgbd_amd import camera, synth
from oracle import cpu_oracle as co
from oracle import kvnet_oracle as ko

# F(2,3)
G2 = torch.tensor([[1.0, 0.0, 0.0], [0.5, 0.5, 0.5], [0.5, -0.5, 0.5], [0.0, 0.0, 1.0]], dtype=torch.float64)
BT2 = torch.tensor([[1.0, 0.0, -1.0, 0.0], [0.0, 1.0, 1.0, 0.0], [0.0, -1.0, 1.0, 0.0], [0.0, 1.0, 0.0, -1.0]], dtype=torch.float64)
AT2 = torch.tensor([[1.0, 1.0, 1.0, 0.0], [0.0, 1.0, -1.0, -1.0]], dtype=torch.float64)
# F(4,3) (Lavin & Gray 2016)
G4 = torch.tensor([[1 / 4, 0, 0], [-1 / 6, -1 / 6, -1 / 6], [-1 / 6, 1 / 6, -1 / 6], [1 / 24, 1 / 12, 1 / 6], [1 / 24, -1 / 12, 1 / 6], [0, 0, 1]], dtype=torch.float64)
BT4 = torch.tensor([[4, 0, -5, 0, 1, 0], [0, -4, -4, 1, 1, 0], [0, 4, -4, -1, 1, 0], [0, -2, -1, 2, 1, 0], [0, 2, -1, -2, 1, 0], [0, 4, 0, -5, 0, 1]], dtype=torch.float64)
AT4 = torch.tensor([[1, 1, 1, 1, 1, 0], [0, 1, -1, 2, -2, 0], [0, 1, 1, 4, 4, 0], [0, 1, -1, 8, -8, 1]], dtype=torch.float64)


def conv3d_wino(x, w, m):
    """x [1,Cin,D,H,W] float32, w [Cout,Cin,3,3,3] -> [1,Cout,D,H,W]: F(m x m, 3x3) in the plane (m = 2 or 4) x F(2,3) along depth,
    every stage in float32 (the weights' transform in float64 then rounded once: the kernels pack U on the GPU in fp32 from exact
    small rationals; a double-rounded U is the favourable case)."""
    G, BT, AT = (G2, BT2, AT2) if m == 2 else (G4, BT4, AT4)
    t = m + 2
    dt = x.dtype
    _, cin, D, H, W = x.shape
    cout = w.shape[0]
    assert D % 2 == 0 and H % m == 0 and W % m == 0
    U = torch.einsum("tz,ay,oczyx,bx->octab", G2, G, w.double(), G).to(dt)                    # [co, ci, 4, t, t]
    xp = F.pad(x[0], (1, 1, 1, 1, 1, 1))                                                       # [ci, D+2, H+2, W+2]
    p = xp.unfold(2, t, m).unfold(3, t, m)                                                     # [ci, D+2, nty, ntx, t(y), t(x)]
    BTf, ATf = BT.to(dt), AT.to(dt)
    V = torch.einsum("ay,cdijyx->cdijax", BTf, p)                                              # rows
    V = torch.einsum("bx,cdijax->cdijab", BTf, V)                                              # columns            [ci, D+2, nty, ntx, t, t]
    Vd = V.unfold(1, 4, 2)                                                                     # [ci, D/2, nty, ntx, t, t, 4(z)]
    Vd = torch.einsum("tz,cpijabz->cptijab", BT2.to(dt), Vd)                                   # depth combine      [ci, D/2, 4, nty, ntx, t, t]
    M = torch.einsum("octab,cptijab->optijab", U, Vd)                                          # channel contraction [co, D/2, 4, nty, ntx, t, t]
    Y = torch.einsum("na,optijab->optijnb", ATf, M)                                            # inverse transform, rows
    Y = torch.einsum("eb,optijnb->optijne", ATf, Y)                                            # columns            [co, D/2, 4, nty, ntx, m, m]
    Y = torch.einsum("kt,optijne->opkijne", AT2.to(dt), Y)                                     # depth fold         [co, D/2, 2, nty, ntx, m, m]
    return Y.permute(0, 1, 2, 3, 5, 4, 6).reshape(1, cout, D, H, W)                            # (p, k) -> z; (i, n) -> y; (j, e) -> x


def knet(sd, vol, conv):
    def cb(x, q):
        return ko._bn(conv(x, sd[q + ".0.weight"]), sd, q + ".1")
    p = "kv_net"
    x = F.relu(cb(vol, p + ".dres0.0"))
    x = F.relu(cb(x, p + ".dres0.2"))
    for i in (1, 2, 3, 4):
        y = F.relu(cb(x, "%s.dres%d.0" % (p, i)))
        x = cb(y, "%s.dres%d.2" % (p, i)) + x
    y = F.relu(cb(x, p + ".classify.0"))
    return F.conv3d(y, sd[p + ".classify.2.weight"], None, 1, 1)


def main():
    H, W, D = 256, 384, 64
    cam = camera.scannet_intrinsics(W // 4, H // 4)
    d_candi = np.linspace(0.1, 5.0, D)
    import neuralrgbd_amd
    model = neuralrgbd_amd.KVNET(64, cam, d_candi, 10.0, 64, None, if_refined=True, refineNet_name="DPV", t_win_r=2)
    sd = synth.seeded_state_dict(model, 0)
    w1, w2 = (synth.noise_window(s, H, W) for s in (101, 102))
    torch.set_num_threads(8)
    with torch.no_grad():
        o1 = ko.step(sd, *w1, cam, d_candi, 10.0, None)
        pred = o1[3]
        ref, src, poses = w2
        BV_cur, feats, full = ko.dnet(sd, ref, src, poses, cam, d_candi, 10.0)
        V = src.shape[1]
        rgb = full[:, -3:]
        KR, Kt = ko._terms(cam, poses[0])
        warped = co.warp_volume(rgb[:V].numpy(), KR, Kt, cam["unit_ray_array_2D"].numpy(), d_candi, cam["intrinsic_M"][0, 2], cam["intrinsic_M"][1, 2])
        h, w = rgb.shape[2:]
        vol = torch.cat((torch.from_numpy(warped).reshape(V * 3, D, h, w), rgb[V][:, None].expand(3, D, h, w), (BV_cur - pred)), 0)[None]
        direct = lambda x, wt: F.conv3d(x, wt, None, 1, 1)
        first = lambda f: (lambda x, wt: f(x, wt) if wt.shape[1] == 64 else F.conv3d(x, wt, None, 1, 1))   # the 16 -> 64 layer stays on its own form
        dpv = {}
        sd64 = {k: v.double() for k, v in sd.items()}
        g64 = knet(sd64, vol.double(), direct)
        dpv["fp64"] = torch.log_softmax(g64[0, 0] + pred[0].double(), dim=0)
        for name, conv in (("direct", direct), ("F2", first(lambda x, wt: conv3d_wino(x, wt, 2))), ("F4", first(lambda x, wt: conv3d_wino(x, wt, 4)))):
            g = knet(sd, vol, conv)
            dpv[name] = torch.log_softmax(g[0, 0] + pred[0], dim=0)
            print("%-6s gain: max|d vs fp64| %.3e mean %.3e" % (name, (g.double() - g64).abs().max().item(), (g.double() - g64).abs().mean().item()))
    print("K-Net of the update frame, config S windows (seeds 101 / 102), all other stages identical (the oracle's):")
    for name in ("direct", "F2", "F4"):
        a = dpv[name]
        e_or = (a - dpv["direct"]).abs()
        e64 = (a.double() - dpv["fp64"]).abs()
        flips = int((a.argmax(0) != dpv["direct"].argmax(0)).sum())
        print("  %-6s DPV vs oracle (fp32 direct): L1 %.3e max %.3e arg-max flips %d   |.. - fp64|: mean %.3e max %.3e"
              % (name, e_or.mean().item(), e_or.max().item(), flips, e64.mean().item(), e64.max().item()))
    r = (dpv["F4"].double() - dpv["fp64"]).abs().mean().item() / (dpv["direct"].double() - dpv["fp64"]).abs().mean().item()
    l1 = (dpv["F4"] - dpv["direct"]).abs().mean().item()
    print("acceptance: DPV L1 vs oracle %.2e (< 5e-5 ?)  |F4 - fp64| / |oracle - fp64| = %.2f (<= 1.25 ?)  ->  %s" %
          (l1, r, "ACCEPT" if (l1 < 5e-5 and r <= 1.25) else "REJECT"))
    # one layer in isolation: relative error of the two forms against float64 direct
    x = torch.relu(torch.randn(1, 64, 16, 32, 32))
    wt = sd["kv_net.dres1.0.0.weight"]
    y64 = F.conv3d(x.double(), wt.double(), None, 1, 1)
    for name, y in (("direct", F.conv3d(x, wt, None, 1, 1)), ("F2", conv3d_wino(x, wt, 2)), ("F4", conv3d_wino(x, wt, 4))):
        e = (y.double() - y64).abs()
        print("  one 64->64 layer, %-6s: max|d|/max|y| %.2e  mean|d|/mean|y| %.2e" % (name, e.max().item() / y64.abs().max().item(), e.mean().item() / y64.abs().mean().item()))


if __name__ == "__main__":
    main()
