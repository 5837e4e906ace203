"""Whole path on the GPU — KVNET.forward + PREDICT through the drop-in surface — vs the CPU oracle
and the golden vectors from the reference's own test()."""
import numpy as np
import pytest
import torch

from conftest import near_tie_mismatches, report
from neuralrgbd_amd import camera, synth
from oracle import gen_golden
from oracle import kvnet_oracle as ko

pytestmark = pytest.mark.gpu


def _model(cam, d_candi, sigma, seed=0):
    import neuralrgbd_amd
    m = neuralrgbd_amd.KVNET(64, cam, d_candi, sigma, 64, None, if_refined=True, refineNet_name="DPV", t_win_r=2)
    sd = synth.seeded_state_dict(m, seed)
    m.load_state_dict(sd)
    return m.cuda(), sd


def _stream(model, cam, d_candi, windows, R_net=False):
    from neuralrgbd_amd.test_step import test
    outs, pred = [], None
    for (r, s, p) in windows:
        Rd = [{"img": r}]
        Sd = [[{"img": s[0, v:v + 1]} for v in range(s.shape[1])]]
        dpv, nxt = test(model, d_candi, [cam], 2, Rd, Sd, p, pred, R_net=R_net)
        outs.append((dpv, nxt))
        pred = nxt
    return outs


def test_two_frame_stream_vs_golden(golden_net):
    """Two frames (first-frame and update branch) through the path against the reference's own outputs."""
    n, g = gen_golden.NET, golden_net
    cam = camera.scannet_intrinsics(n["W"] // 4, n["H"] // 4)
    d_candi = np.linspace(n["d_min"], n["d_max"], n["D"])
    model, _ = _model(cam, d_candi, n["sigma"], n["weight_seed"])
    windows = [synth.noise_window(s, n["H"], n["W"]) for s in n["seeds"]]
    (bv1, p1), (dpv2, p2) = _stream(model, cam, d_candi, windows)
    refined = _stream(model, cam, d_candi, windows, R_net=True)[1][0]
    res = {}
    for name, got, want in (("BV_cur f1", bv1, g["bv_cur_f1"]), ("BV_predict f1", p1, g["pred_f1"]),
                            ("DPV f2", dpv2, g["dpv_f2"]), ("BV_predict f2", p2, g["pred_f2"])):
        got = got[0].cpu().numpy()
        res[name] = report("GPU path " + name + " vs reference", got, want) + (near_tie_mismatches(got, want, 1e-3),)
    # the contract (BASELINE.json): L1 < 1e-4 on every volume, arg-max depth index identical (res[.][2] = raw
    # mismatch count, no near-tie allowance); max-abs is printed — tests/test_gpu_parity_configs.py::test_fp64_yardstick
    # shows the reference's own output is ~2e-3 max away from exact arithmetic after the K-Net
    for name in res:
        assert res[name][1] < 1e-4, (name, res[name])
        if "predict" not in name:   # BV_predict: faces overwritten with a constant => its arg-max is a tie, not a depth
            assert res[name][2] == 0, (name, res[name])
    assert res["BV_cur f1"][0] < 2e-3
    sub = refined[0, :, ::4, ::4].cpu().numpy()
    _, r_mean, r_mism = report("GPU path R(DPV) f2 vs reference", sub, g["refined_f2_sub"])
    assert r_mean < 1e-4 and r_mism == 0


def test_update_frame_vs_cpu_oracle_config_S_small():
    """One update-branch frame at a second shape/seed against the oracle run on this machine's CPU."""
    H, W, D = 256, 256, 24
    cam = camera.scannet_intrinsics(W // 4, H // 4)
    d_candi = np.linspace(0.1, 5, D)
    model, sd = _model(cam, d_candi, 10.0, seed=1)
    w1, w2 = synth.noise_window(31, H, W), synth.noise_window(32, H, W)
    (bv1, p1), (dpv2, p2) = _stream(model, cam, d_candi, [w1, w2])
    o1 = ko.step(sd, *w1, cam, d_candi, 10.0, None)
    o2 = ko.step(sd, *w2, cam, d_candi, 10.0, o1[3])
    a = report("GPU vs oracle BV_cur", bv1[0].cpu().numpy(), o1[2][0].numpy())
    b = report("GPU vs oracle DPV", dpv2[0].cpu().numpy(), o2[1][0].numpy())
    c = report("GPU vs oracle BV_predict", p2[0].cpu().numpy(), o2[3][0].numpy())
    assert a[1] < 1e-4 and b[1] < 1e-4 and c[1] < 1e-4
    assert a[2] == 0 and b[2] == 0


def test_rendered_scene_vs_golden(golden_scene):
    s, g = gen_golden.SCENE, golden_scene
    cam = camera.scannet_intrinsics(s["W"] // 4, s["H"] // 4)
    cam_full = camera.scannet_intrinsics(s["W"], s["H"])
    d_candi = np.linspace(0.1, 5, s["D"])
    model, _ = _model(cam, d_candi, s["sigma"], 0)
    r, sr, p, depth = synth.rendered_window(s["seed"], s["H"], s["W"], cam_full)
    with torch.no_grad():
        bv, _ = model.d_net(r.cuda(), sr.cuda(), p.cuda())
    got = bv[0].cpu().numpy()
    mx, mean, mism = report("GPU D-Net on rendered scene vs reference", got, g["bv_cur"])
    assert mean < 1e-4 and mism == 0


def test_first_frame_and_invalid_state_fallbacks():
    """KVNET.py:138-143: no BV_predict, or a NaN-flagged one, returns the D-Net pair twice."""
    H, W, D = 256, 256, 8
    cam = camera.scannet_intrinsics(W // 4, H // 4)
    d_candi = np.linspace(0.1, 5, D)
    model, _ = _model(cam, d_candi, 10.0)
    r, s, p = synth.noise_window(41, H, W)
    with torch.no_grad():
        out = model(r.cuda(), s.cuda(), p.cuda(), torch.zeros(1), cam_intrinsics=[cam], BV_predict=None)
        assert out[0] is out[1] and out[2] is out[3]
        assert out[0].shape == (1, D, H, W) and out[2].shape == (1, D, H // 4, W // 4)
        bad = torch.full((1, D, H // 4, W // 4), float("nan"), device="cuda")
        out2 = model(r.cuda(), s.cuda(), p.cuda(), torch.zeros(1), cam_intrinsics=[cam], BV_predict=bad)
        assert out2[2] is out2[3]
