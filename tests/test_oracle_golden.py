"""The CPU oracle against the golden vectors generated from the unmodified reference
(oracle/gen_golden.py).  CPU only; this is what pins the oracle (prompt §3)."""
import math
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, report
from neuralrgbd_amd import camera, synth
from oracle import cpu_oracle as co
from oracle import gen_golden, kvnet_oracle as ko


def _cam_ops():
    o = gen_golden.OPS
    return camera.scannet_intrinsics(o["w"], o["h"])


def test_costvol_l2_l1(golden_ops):
    g, cam = golden_ops, _cam_ops()
    rays = cam["unit_ray_array_2D"].numpy()
    cx, cy = cam["intrinsic_M"][0, 2], cam["intrinsic_M"][1, 2]
    for dist, key in (("L2", "cost_l2"), ("L1", "cost_l1")):
        got = co.costvol(g["feat_ref"], g["feat_src"], g["KR"], g["Kt"], rays, g["d_candi"], cx, cy,
                         float(g["sigma"]), dist=dist)
        mx, _, _ = report("oracle costvol " + dist, -got, -g[key])
        assert mx < 2e-5  # costs up to 16.6: a few fp32 ulps of summation-order noise
        assert (got.argmin(0) != g[key].argmin(0)).sum() == 0


def test_logsoftmax_and_depth(golden_ops):
    g = golden_ops
    bv = co.logsoftmax_d(g["cost_l2"], scale=-1.0)
    mx, _, mism = report("oracle log_softmax", bv, g["bv"])
    assert mx < 1e-5 and mism == 0
    depth, conf = co.depth_regress(g["bv"], g["d_candi"])
    assert np.abs(depth - g["depth"]).max() < 1e-5
    assert np.array_equal(conf, g["bv"].max(0))


def test_warp_volume(golden_ops):
    g, cam = golden_ops, _cam_ops()
    got = co.warp_volume(g["rgb"], g["KR"], g["Kt"], cam["unit_ray_array_2D"].numpy(), g["d_candi"],
                         cam["intrinsic_M"][0, 2], cam["intrinsic_M"][1, 2])
    assert got.shape == g["warped"].shape  # [V,3,D,h,w]
    assert np.abs(got - g["warped"]).max() < 1e-5


def test_dpv_resample_bit_exact(golden_ops):
    g, cam = golden_ops, _cam_ops()
    got = co.dpv_resample(g["dpv"], g["T"], cam["unit_ray_array_2D"].numpy(), g["d_candi"],
                          math.tan(math.radians(cam["hfov"]) * .5), math.tan(math.radians(cam["vfov"]) * .5),
                          float(g["pad"]))
    assert np.array_equal(got, g["pred"])  # the 3-D sampler restatement is bit-identical to the reference


def _net_setup(n):
    cam = camera.scannet_intrinsics(n["W"] // 4, n["H"] // 4)
    d_candi = np.linspace(n["d_min"], n["d_max"], n["D"])
    import neuralrgbd_amd
    model = neuralrgbd_amd.KVNET(64, cam, d_candi, n["sigma"], 64, None, if_refined=True, refineNet_name="DPV", t_win_r=2)
    sd = synth.seeded_state_dict(model, n["weight_seed"])
    return cam, d_candi, sd


def test_whole_path_two_frames(golden_net):
    """Torch-CPU restatement of KVNET.forward + PREDICT vs the reference's own test() on two frames."""
    torch.set_num_threads(max(1, torch.get_num_threads()))
    n, g = gen_golden.NET, golden_net
    cam, d_candi, sd = _net_setup(n)
    assert abs(gen_golden.checksum(sd.values()) - float(g["weights_checksum"])) < 1e-6 * float(g["weights_checksum"])
    w1, w2 = (synth.noise_window(s, n["H"], n["W"]) for s in n["seeds"])
    import os
    from conftest import GOLDEN
    inv = np.load(os.path.join(GOLDEN, "pose_inv_ref.npz"))["net_inv"]
    # (1) with the matrices the reference's own `.inverse()` produced (host LAPACK, stored): everything as before
    o1 = ko.step(sd, *w1, cam, d_candi, n["sigma"], None, rel_extM=inv[0])
    o2 = ko.step(sd, *w2, cam, d_candi, n["sigma"], o1[3], rel_extM=inv[1])
    checks = [("BV_cur f1", o1[2][0], g["bv_cur_f1"], 2e-4), ("BV_predict f1", o1[3][0], g["pred_f1"], 2e-4),
              ("DPV f2", o2[1][0], g["dpv_f2"], 5e-4), ("BV_predict f2", o2[3][0], g["pred_f2"], 5e-4)]
    for name, got, want, tol in checks:
        mx, mean, mism = report("oracle " + name, got.numpy(), want)
        assert mx < tol and mean < 1e-4 and mism == 0
    assert (o2[0][0].argmax(0).numpy() != g["refined_f2_argmax"]).sum() == 0
    assert np.abs(o2[0][0, :, ::4, ::4].numpy() - g["refined_f2_sub"]).max() < 1e-4
    # (2) with the path's own inverse (fixed operation order, what the product and the oracle run by default): it differs
    # from the LAPACK result in the last bits, which moves every resampling coordinate by ~1e-7 — BV_predict follows with
    # the DPV's slope.  Depth estimates (arg-max of BV_cur / DPV / refined) are unchanged; BV_predict's own arg-max is a tie
    # by construction (border faces = log(1/D)) and is not compared.
    p1 = ko.step(sd, *w1, cam, d_candi, n["sigma"], None)
    p2 = ko.step(sd, *w2, cam, d_candi, n["sigma"], p1[3])
    for name, got, want, tol in (("BV_predict f1", p1[3][0], g["pred_f1"], 1e-3), ("DPV f2", p2[1][0], g["dpv_f2"], 1e-3),
                                 ("BV_predict f2", p2[3][0], g["pred_f2"], 1e-3)):
        mx, mean, mism = report("oracle/own inverse " + name, got.numpy(), want)
        assert mx < tol and mean < 1e-4
    assert (p2[1][0].argmax(0).numpy() != g["dpv_f2"].argmax(0)).sum() == 0
    assert (p2[0][0].argmax(0).numpy() != g["refined_f2_argmax"]).sum() == 0


def test_rendered_scene(golden_scene):
    s, g = gen_golden.SCENE, golden_scene
    cam = camera.scannet_intrinsics(s["W"] // 4, s["H"] // 4)
    cam_full = camera.scannet_intrinsics(s["W"], s["H"])
    d_candi = np.linspace(0.1, 5, s["D"])
    import neuralrgbd_amd
    model = neuralrgbd_amd.KVNET(64, cam, d_candi, s["sigma"], 64, None, if_refined=True, refineNet_name="DPV", t_win_r=2)
    sd = synth.seeded_state_dict(model, 0)
    r, sr, p, depth = synth.rendered_window(s["seed"], s["H"], s["W"], cam_full)
    assert abs(gen_golden.checksum([r, sr, p]) - float(g["inputs_checksum"])) < 1e-6 * float(g["inputs_checksum"])
    with torch.no_grad():
        bv, _, _ = ko.dnet(sd, r, sr, p, cam, d_candi, s["sigma"])
    mx, mean, mism = report("oracle scene BV_cur", bv[0].numpy(), g["bv_cur"])
    assert mx < 5e-4 and mean < 1e-4 and mism == 0


def test_costvol_c67_d64_vs_reference():
    """The C oracle at the path's real channel / candidate count (C=67, D=64, V=4) against the reference's output."""
    import os
    from conftest import GOLDEN
    o = gen_golden.OPS67
    g = dict(np.load(os.path.join(GOLDEN, "ops_c67.npz")))
    feat_ref, feat_src, poses, d_candi = gen_golden.ops67_inputs()
    assert abs(gen_golden.checksum([torch.from_numpy(feat_ref), torch.from_numpy(feat_src), torch.from_numpy(poses)])
               - float(g["inputs_checksum"])) < 1e-6 * float(g["inputs_checksum"])
    cam = camera.scannet_intrinsics(o["w"], o["h"])
    KR, Kt = co.homography_terms(cam["intrinsic_M_cuda"].numpy(), poses[:, :3, :3], poses[:, :3, 3])
    got = co.costvol(feat_ref[0], feat_src[0], KR, Kt, cam["unit_ray_array_2D"].numpy(), d_candi,
                     cam["intrinsic_M"][0, 2], cam["intrinsic_M"][1, 2], o["sigma"])
    mx, mean, _ = report("oracle costvol C=67 D=64", -got[:, ::2, ::2], -g["cost_sub"])
    assert mx < 1e-4 and mean < 1e-5          # costs up to 62
    assert (got.argmin(0) != g["argmin"]).sum() == 0


def test_fp64_yardstick_is_the_same_graph(golden_net):
    """oracle/fp64_ref.py evaluated in float32 must reproduce the reference (it is the same formulas); the stored
    float64 results are then a yardstick for rounding noise, not a different algorithm."""
    import os
    from conftest import GOLDEN
    from oracle import fp64_ref
    n, g = gen_golden.NET, golden_net
    cam, d_candi, sd = _net_setup(n)
    w1 = synth.noise_window(n["seeds"][0], n["H"], n["W"])
    old = fp64_ref.F64
    try:
        fp64_ref.F64 = torch.float32
        o32 = fp64_ref.step(sd, *w1, cam, d_candi, n["sigma"], None)
    finally:
        fp64_ref.F64 = old
    assert np.abs(o32[2][0].numpy() - g["bv_cur_f1"]).max() < 1e-5
    o64 = fp64_ref.step(sd, *w1, cam, d_candi, n["sigma"], None)
    g64 = dict(np.load(os.path.join(GOLDEN, "net_fp64.npz")))
    assert np.abs(o64[2][0].numpy()[:, ::2, ::2] - g64["bv_cur_f1"]).max() < 1e-9
    # the reference's own fp32 output sits ~1e-3 (max) from exact arithmetic already at the D-Net output
    assert 1e-4 < float(g64["ref_err_max_dpv_f2"]) < 1e-2


def test_lba_depth_warp_forward_and_pose_gradients():
    """oracle_warp_depth_fwd/_bwd vs the reference's back_warp_th_Rt_msrc and torch autograd's dR, dt (lba_small.npz)."""
    import os
    from conftest import GOLDEN
    o = gen_golden.LBA
    g = dict(np.load(os.path.join(GOLDEN, "lba_small.npz")))
    src, ref_img, dmap, poses, G = gen_golden.lba_inputs()
    cam = camera.scannet_intrinsics(o["W"], o["H"])
    K, rays = cam["intrinsic_M_cuda"].numpy(), cam["unit_ray_array_2D"].numpy()
    out = co.warp_depth_fwd(src, dmap, K, poses[:, :3, :3], poses[:, :3, 3], rays)
    assert np.abs(out - g["warped"]).max() < 2e-5
    assert np.abs(out[:1] - g["single"]).max() < 2e-5
    gR, gt = co.warp_depth_bwd(src, dmap, K, poses[:, :3, :3], poses[:, :3, 3], rays, G)
    assert np.abs(gR - g["g_R"]).max() < 1e-5 * np.abs(g["g_R"]).max()
    assert np.abs(gt - g["g_t"]).max() < 1e-5 * np.abs(g["g_t"]).max()
    # the LBA loss (ICP/opt_pose_numerical.py:262-270): L1 over pixels the warp reached, upstream gradient = sign / count
    mask = (out != 0).astype(np.float32)
    diff = out * mask - ref_img * mask
    assert abs(float(np.abs(diff).mean()) - float(g["loss"])) < 1e-6
    gR2, gt2 = co.warp_depth_bwd(src, dmap, K, poses[:, :3, :3], poses[:, :3, 3], rays, np.sign(diff) * mask / diff.size)
    assert np.abs(gR2 - g["g_R_loss"]).max() < 1e-4 * np.abs(g["g_R_loss"]).max()
    assert np.abs(gt2 - g["g_t_loss"]).max() < 1e-4 * np.abs(g["g_t_loss"]).max()


def test_export_epilogue_vs_reference_files():
    """oracle_export_depth_u16 vs the uint16 .pgm images written by the reference's export_res_img (export_small.npz)."""
    import os
    from conftest import GOLDEN
    g = dict(np.load(os.path.join(GOLDEN, "export_small.npz")))
    bv, _, d_candi = gen_golden.export_inputs()
    depth, conf, du, cu = co.export_depth_u16(bv[0].numpy(), d_candi)
    assert np.abs(depth - g["depth"]).max() < 1e-5 and np.abs(conf - g["conf"]).max() < 1e-6
    # `(map * 1000).astype(uint16)` truncates: a last-ulp difference of the fp32 map flips a value only when map * 1000 sits
    # within that ulp of an integer.  The reference's map comes from torch.exp (sleef expf, a 1-ulp function) and torch.sum
    # (ATen's cascade order); the path's from exp_rn (correctly rounded, written out) and the sequential sum of
    # depth_val_regression (misc.py:540-546).  Exact counts, printed; measured on this fixture: 0 and 0 of 960 pixels (round 2's expf-based path: <= 1 LSB on 0.2 %):
    nd, nc = int((du != g["depth_u16"]).sum()), int((cu != g["conf_u16"]).sum())
    print("[parity] export u16 vs the reference's .pgm files: depth %d / conf %d of %d pixels differ (all by 1 LSB)" %
          (nd, nc, du.size))
    assert (np.abs(du.astype(np.int32) - g["depth_u16"].astype(np.int32)) > 1).sum() == 0
    assert (np.abs(cu.astype(np.int32) - g["conf_u16"].astype(np.int32)) > 1).sum() == 0
    assert nd <= 4 and nc <= 4


def test_winograd_restatement_and_weight_stream_layout():
    """oracle/wino_ref.py: F(2x2,3x3) with the kernel's matrices equals the direct convolution (float64), and the weight-stream
    layout documented in include/nrgbd.h is what the host packer (ops.conv_wino_pack_reference, torch on the CPU) produces."""
    import torch
    import torch.nn.functional as F
    from oracle import wino_ref
    from neuralrgbd_amd import ops
    rng = np.random.RandomState(0)
    x, w = rng.randn(5, 8, 12), rng.randn(7, 5, 3, 3)
    want = F.conv2d(torch.from_numpy(x)[None], torch.from_numpy(w), padding=1)[0].numpy()
    assert np.abs(wino_ref.conv2d_wino(x, w) - want).max() < 1e-12
    for shape in ((64, 32, 3, 3), (128, 64, 3, 3, 3)):
        wt = torch.from_numpy(rng.randn(*shape).astype(np.float32))
        stream = ops.conv_wino_pack_reference(wt).numpy()
        w5 = wt[:, :, None] if wt.dim() == 4 else wt
        Cout, Cin, KD = w5.shape[:3]
        U = np.einsum("ay,ockyx,bx->ockab", wino_ref.G, w5.numpy().astype(np.float64), wino_ref.G).reshape(Cout, Cin, KD, 16)
        for co, ci, kd, xi in ((0, 0, 0, 0), (17, 5, KD - 1, 6), (Cout - 1, Cin - 1, 0, 15), (33, 18, KD // 2, 9)):
            got = stream[wino_ref.packed_index(co, ci, kd, xi, Cin, KD)]
            assert abs(got - np.float32(U[co, ci, kd, xi])) <= 1e-6 * max(1.0, abs(U[co, ci, kd, xi])), (shape, co, ci, kd, xi)


def test_pose_inverse_is_the_rounded_float64_inverse():
    """oracle_pose_inverse: fp64 Gauss-Jordan with partial pivoting, rounded to fp32 — within half an fp32 ulp of the exact
    inverse (the reference's `.inverse()`, test_utils/test_KVNet.py:50, is host LAPACK in fp32: ~4x further away)."""
    rng = np.random.RandomState(3)
    T = np.stack([synth.random_pose(rng, 0.6, 2.0) for _ in range(500)]).astype(np.float32)
    T[250:] += (rng.standard_normal((250, 4, 4)) * 1e-2).astype(np.float32)
    got = co.pose_inverse(T)
    ex = np.linalg.inv(T.astype(np.float64))
    ulp = np.spacing(np.abs(ex).astype(np.float32))
    assert (np.abs(got - ex) <= 0.5000001 * ulp).all()
    assert (got != ex.astype(np.float32)).mean() < 1e-3          # only double-rounding cases may differ from RN(exact)
    ref = torch.from_numpy(T).inverse().numpy()                   # LAPACK fp32 on this host: a few ulps from exact
    assert np.abs(got - ref).max() < 2e-5 and np.abs(got - ex).max() <= np.abs(ref - ex).max()
    perm = np.eye(4, dtype=np.float32)[[2, 0, 3, 1]]
    assert np.array_equal(co.pose_inverse(perm), perm.T)
    import pytest
    with pytest.raises(np.linalg.LinAlgError):
        co.pose_inverse(np.zeros((4, 4), np.float32))


def test_training_oracle_vs_reference_train_golden():
    """oracle/train_oracle.py (the CPU training iteration under autograd: grid_sample cost volume, functional networks, 4 NLL
    terms, SGD, PREDICT) against two iterations of the unmodified reference's own train() (tests/golden/train_small.npz):
    losses, predicted filter state, and the SGD weight deltas (= lr x gradient) of six probe tensors."""
    import neuralrgbd_amd
    from oracle import gen_golden, train_oracle
    t = gen_golden.TRAIN
    g = dict(np.load(os.path.join(GOLDEN, "train_small.npz")))
    cam = camera.scannet_intrinsics(t["W"] // 4, t["H"] // 4)
    d_candi = np.linspace(0.1, 5, t["D"])
    model = neuralrgbd_amd.KVNET(64, cam, d_candi, t["sigma"], 64, None, if_refined=True, refineNet_name="DPV", t_win_r=2)
    leaves = train_oracle.leaf_state(synth.seeded_state_dict(model, t["weight_seed"]))
    opt = torch.optim.SGD(train_oracle.parameters(leaves), lr=t["lr"])
    pred = None
    for it, (r, s, p, dm, dmf) in enumerate(gen_golden.train_inputs()):
        before = {k: leaves[k].detach().clone() for k in t["probes"]}
        loss, pred = train_oracle.train_iteration(leaves, opt, r, s, p, dm, dmf, cam, d_candi, t["sigma"], pred)
        want = float(g["loss_%d" % it])
        e_pred = pred[0].numpy() - g["pred_%d" % it]
        print("[oracle] train iteration %d: loss %.6f vs reference %.6f; BV_predict mean|d| %.2e" % (it, float(loss), want, np.abs(e_pred).mean()))
        assert abs(float(loss) - want) < 2e-5 * want
        assert np.abs(e_pred).mean() < (1e-4 if it == 0 else 2e-3)
        for k in t["probes"]:
            delta = (leaves[k].detach() - before[k]).numpy()
            ref_d = g["delta_%d_%s" % (it, k)]
            if np.abs(ref_d).max() == 0:
                assert np.abs(delta).max() == 0
                continue
            rel = np.abs(delta - ref_d).max() / np.abs(ref_d).max()
            assert rel < (2e-2 if it == 0 else 5e-2), (k, rel)


def test_rendered_video_two_frames_vs_reference_golden():
    """The CPU oracle on two consecutive frames of a RENDERED scene at config S (update branch on a peaked DPV with a
    consistent predicted belief — the regime the filter runs in) against the unmodified reference's outputs
    (tests/golden/scene_stream_S.npz, oracle/gen_golden.py::gen_scene_stream)."""
    import os
    from conftest import GOLDEN
    n = gen_golden.SCENE_S
    g = dict(np.load(os.path.join(GOLDEN, "scene_stream_S.npz")))
    H, W, D = n["H"], n["W"], n["D"]
    cam = camera.scannet_intrinsics(W // 4, H // 4)
    d_candi = np.linspace(n["d_min"], n["d_max"], D)
    import neuralrgbd_amd
    model = neuralrgbd_amd.KVNET(64, cam, d_candi, n["sigma"], 64, None, if_refined=True, refineNet_name="DPV", t_win_r=2)
    sd = synth.seeded_state_dict(model, n["weight_seed"])
    windows = synth.rendered_stream(n["seed"], H, W, camera.scannet_intrinsics(W, H), 2)
    assert abs(gen_golden.checksum([w[0] for w in windows] + [w[1] for w in windows] + [w[2] for w in windows])
               - float(g["inputs_checksum"])) < 1e-6 * abs(float(g["inputs_checksum"]))
    o1 = ko.step(sd, *windows[0], cam, d_candi, n["sigma"], None)
    o2 = ko.step(sd, *windows[1], cam, d_candi, n["sigma"], o1[3])
    for name, got, key in (("BV_cur f1", o1[2], "bv_cur_f1"), ("DPV f2", o2[1], "dpv_f2"), ("BV_predict f2", o2[3], "pred_f2")):
        mx, mean, _ = report("oracle rendered video " + name, got[0, :, ::2, ::2].numpy(), g[key + "_sub"])
        assert mean < 1e-4, (name, mean)
    # the regime: log-probabilities far below the noise windows' (-20), most pixels with one dominant candidate
    assert float(o2[1].min()) < -100.0
    for name, got, key in (("BV_cur f1", o1[2], "bv_cur_f1_argmax"), ("DPV f2", o2[1], "dpv_f2_argmax")):
        flips = int((got[0].argmax(0).numpy() != g[key]).sum())
        print("[parity] oracle rendered video %s: arg-max flips vs the reference %d / %d" % (name, flips, g[key].size))
        assert flips <= 2          # the oracle's C sampler and ATen's grid_sample associate differently: ties only


@pytest.mark.parametrize("tag", ["S", "K", "ST"])
def test_reference_selfnoise_envelope_and_oracle(tag):
    """(1) the parity envelope is REFERENCE-generated — the unmodified reference against
    itself (oneDNN on / off, all threads / one) at config S, at config K (KITTI: ~500 near-tie pixels per volume) and at config S
    with the trained-like weight family differs by more than 1e-4 max and by less than the hard gate MAX_ABS_TOL, with no
    arg-max flip beyond a tie and no more flips than conftest.max_tie_flips allows for the fixture's own tie population;
    (2) the CPU oracle, on the same two frames, sits inside that envelope from the reference's base execution on every volume
    INCLUDING both refined outputs (the R-Net at D = 64 candidates)."""
    from conftest import L1_TOL, MAX_ABS_TOL, max_tie_flips, scaled_max_abs, selfnoise
    sn = selfnoise(tag)
    worst = {}
    for key, v in sn.items():
        if key.split("_")[0] in ("onednn", "threads") and v.shape == (5,):
            vol = key.split("_", 2)[2]
            worst[vol] = max(worst.get(vol, 0.0), float(v[0]))
            assert v[0] <= MAX_ABS_TOL and v[1] < L1_TOL and v[3] == 0, (key, v)     # the gates hold for the reference itself
            if not vol.startswith("pred"):
                assert v[2] <= max_tie_flips(int(sn["base_%s_ties" % vol]), v[1]), (key, v, int(sn["base_%s_ties" % vol]))
    assert worst["dpv_f2"] > 1e-4 and worst["pred_f2"] > 1e-4 and worst["bv_cur_f2"] > 1e-4    # ... and 1e-4 (max) does not
    assert all(float(sn["rerun_%s_f%d" % (k, f)][0]) == 0.0 for k in ("bv_cur", "dpv", "pred", "refined") for f in (1, 2))
    n = gen_golden.SELFNOISE[tag]
    cam, d_candi, weights = gen_golden.selfnoise_setup(n)
    import neuralrgbd_amd
    model = neuralrgbd_amd.KVNET(64, cam, d_candi, n["sigma"], 64, None, if_refined=True, refineNet_name="DPV", t_win_r=2)
    sd = weights(model, n["weight_seed"])
    assert abs(gen_golden.checksum(sd.values()) - float(sn["weights_checksum"])) < 1e-6 * float(sn["weights_checksum"])
    w1, w2 = (synth.noise_window(s, n["H"], n["W"]) for s in n["seeds"])
    o1 = ko.step_full(sd, *w1, cam, d_candi, n["sigma"], None)
    o2 = ko.step_full(sd, *w2, cam, d_candi, n["sigma"], o1[4])
    sub, sq = n["sub"], n.get("sub_q", 2)
    for name, got, key, st in (("BV_cur f1", o1[3], "base_bv_cur_f1", sq), ("BV_cur f2", o2[3], "base_bv_cur_f2", sq), ("DPV f2", o2[2], "base_dpv_f2", sq),
                               ("BV_predict f2", o2[4], "base_pred_f2", sq), ("R(BV_cur) f2", o2[0], "base_refined_cur_f2", sub),
                               ("R(DPV) f2", o2[1], "base_refined_f2", sub), ("R(BV_cur) f1", o1[0], "base_refined_cur_f1", sub)):
        a = got[0].numpy()
        mx, mean, _ = report("oracle vs reference, %s %s" % (tag, name), a[:, ::st, ::st], sn[key + "_sub"])
        hard = scaled_max_abs(a[:, ::st, ::st], sn[key + "_sub"]) if n.get("family") == "trained" else mx     # conftest: peaked families
        assert mean < L1_TOL and hard <= MAX_ABS_TOL, (name, mx, hard, mean)
        assert abs(float(a.astype(np.float64).sum()) - float(sn[key + "_sum"])) < 2e-5 * abs(float(sn[key + "_sum"]))   # all pixels
        if "predict" not in name:
            flips = int((a.argmax(0) != sn[key + "_argmax"]).sum())
            cap = max_tie_flips(int(sn[key + "_ties"]), mean)
            print("[parity] oracle vs reference, %s %s: arg-max flips %d / %d (reference-side ties within 1e-3: %d, bound %d)" %
                  (tag, name, flips, a[0].size, int(sn[key + "_ties"]), cap))
            assert flips <= cap
