"""CPU-side checks: the C-ABI library loads and exports every symbol include/nrgbd.h declares, the
host mirror keeps the reference's contracts, and the product path refuses to run without a GPU."""
import ctypes
import json
import os
import sys
import re

import numpy as np
import pytest
import torch

from conftest import GOLDEN, ROOT
from neuralrgbd_amd import camera, synth


def _header_functions():
    text = open(os.path.join(ROOT, "include", "nrgbd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(nrgbd_[a-z0-9_]+)\s*\(", text)))


def test_library_builds_loads_and_exports_the_header():
    from neuralrgbd_amd import _lib, build
    path = build.build()
    assert os.path.isfile(path)
    lib = ctypes.CDLL(path)
    names = _header_functions()
    assert len(names) >= 8
    for name in names:
        assert hasattr(lib, name), "libnrgbd_hip.so does not export %s" % name
    assert sorted(_lib.SIGNATURES) == names  # the ctypes table covers exactly the header
    assert b"gfx950" in _lib.load().nrgbd_version()
    assert _lib.load().nrgbd_strerror(-2).startswith(b"a dimension")


def test_no_cpu_fallback():
    from neuralrgbd_amd import _lib, ops
    with pytest.raises(_lib.NrgbdError):
        ops.pack_nhwc(torch.zeros(1, 4, 8, 8))
    with pytest.raises(_lib.NrgbdError):
        ops.dpv_resample(torch.zeros(4, 8, 8), torch.eye(4), torch.zeros(3, 64), torch.zeros(4), 1, 1, 1, 1, 0)
    # the round-2 entry points fail the same way (and the shape errors of the C side need no GPU to be reported)
    with pytest.raises(_lib.NrgbdError):
        ops.conv_wino(torch.zeros(2, 8, 16, 64), torch.zeros(64 * 64 * 16), 64, 1)
    with pytest.raises(_lib.NrgbdError):
        ops.conv_wino_pack(torch.zeros(64, 64, 3, 3))
    with pytest.raises(_lib.NrgbdError):
        ops.conv2d_taps(torch.zeros(1, 8, 8, 16), torch.zeros(16 * 32), 32, 1)
    lib = _lib.load()
    assert lib.nrgbd_conv_wino_tiles(5, 192, 256, 1) == 5 * 24 * 16 and lib.nrgbd_conv_wino_tiles(5, 192, 256, 2) == 5 * 12 * 8 * 4
    assert lib.nrgbd_conv_wino_tiles(1, 8, 16, 3) < 0                                     # dilation 3 is not a form of the kernel
    assert lib.nrgbd_conv2d_wgrad_workgroups(5, 64, 96, 64, 64) == 240                    # one 8x16 pixel tile per workgroup
    assert lib.nrgbd_conv2d_wgrad_workgroups(5, 192, 256, 320, 128) == 182                # 1,920 tiles; 10 weight blocks share ~256 MB of partials
    assert lib.nrgbd_conv2d_wgrad_workgroups(5, 192, 256, 64, 64) == 1024                 # one block: up to 1,024 workgroups
    assert lib.nrgbd_conv_wino_f32(None, None, 0, None, None, 0, None, None, None, None, 1, 8, 16, 64, 64, 1, 1, None) < 0   # NULL


def test_product_package_does_not_import_the_oracle():
    pkg = os.path.join(ROOT, "neuralrgbd_amd")
    for fn in os.listdir(pkg):
        if fn.endswith(".py"):
            src = open(os.path.join(pkg, fn)).read()
            assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), fn
            assert "ref_shim" not in src and "/root/reference" not in src, fn


def test_state_dict_contract_against_reference_keys():
    import neuralrgbd_amd
    want = {k: tuple(v) for k, v in json.load(open(os.path.join(GOLDEN, "state_keys.json"))).items()}
    n_d = 16  # the golden key list was taken with D=16 (R-Net widths depend on D)
    cam = camera.scannet_intrinsics(80, 64)
    model = neuralrgbd_amd.KVNET(64, cam, np.linspace(.1, 5, n_d), 10., 64, None, if_refined=True,
                                 refineNet_name="DPV", t_win_r=2)
    got = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    assert got == want and len(got) == 459
    # the shared feature CNN appears under both prefixes with the same storage
    sd = model.state_dict()
    a = sd["feature_extractor.feature_extraction.firstconv.0.0.weight"]
    b = sd["d_net.feature_extraction.feature_extraction.firstconv.0.0.weight"]
    assert a.data_ptr() == b.data_ptr()
    # DataParallel-style checkpoints ('module.' prefix) load after stripping, like utils/models.py:39-59
    ck = {"module." + k: v for k, v in synth.seeded_state_dict(model, 3).items()}
    model.load_state_dict({k[len("module."):]: v for k, v in ck.items()})
    # feature CNN 3,343,648 and K-Net 1,136,704 parameters at the canonical flags
    assert sum(p.numel() for p in model.feature_extractor.parameters()) == 3343648
    assert sum(p.numel() for p in model.kv_net.parameters()) == 1136704
    full = neuralrgbd_amd.KVNET(64, cam, np.linspace(.1, 5, 64), 10., 64, None)
    assert sum(p.numel() for p in full.parameters()) == 5287156


def test_camera_dict_schema_and_convention():
    cam = camera.scannet_intrinsics(96, 64)
    assert set(cam) == {"hfov", "vfov", "unit_ray_array", "unit_ray_array_2D", "intrinsic_M_cuda",
                        "focal_length", "intrinsic_M"}
    assert cam["unit_ray_array"].shape == (64, 96, 3) and cam["unit_ray_array"].dtype == np.float64
    assert cam["unit_ray_array_2D"].shape == (3, 64 * 96) and cam["unit_ray_array_2D"].dtype == torch.float32
    assert cam["intrinsic_M"].shape == (3, 4) and cam["intrinsic_M"][0, 2] == 48.0 and cam["intrinsic_M"][1, 2] == 32.0
    # (u - cx)/cx of the pixel centre equals 2(x+.5)/w - 1: K projects ray -> pixel centre + 0.5
    K = cam["intrinsic_M"][:3, :3]
    uv = K @ cam["unit_ray_array"][10, 20]
    assert abs(uv[0] - 20.5) < 1e-9 and abs(uv[1] - 10.5) < 1e-9


def test_synth_is_deterministic():
    a = synth.noise_window(5, 8, 12)
    b = synth.noise_window(5, 8, 12)
    assert all(torch.equal(x, y) for x, y in zip(a, b))
    R = a[2][0, :, :3, :3].double()
    assert torch.allclose(R @ R.transpose(1, 2), torch.eye(3, dtype=torch.float64).expand(4, 3, 3), atol=1e-6)


def test_spp_pooling_rewrite_matches_avg_pool():
    """The training-path rewrite of the SPP pooling (nets.PSMFeatures: crop + reshape + mean) equals avg_pool2d.  (The bilinear
    up-sampling of the branches is a kernel pair now: tests/test_gpu_cnn.py::test_upsample_bilinear_align_corners_forward_and_adjoint.)"""
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(0)
    d = torch.randn(2, 3, 64, 96, generator=g)
    for k in (64, 32, 16, 8):
        ph, pw = 64 // k, 96 // k
        a = d[:, :, :ph * k, :pw * k].reshape(2, 3, ph, k, pw, k).mean((3, 5))
        b = F.avg_pool2d(d, (k, k), stride=(k, k))
        assert a.shape == b.shape and (a - b).abs().max().item() < 1e-6


def test_division_by_the_principal_point_is_exact():
    """csrc/common.hpp::div_by_const replaces the reference's `(u - cx) / cx` (homography.py:441-445) by a multiply and two
    fused multiply-adds with rc = RN(1/cx): it must be the SAME fp32 number for every dividend, or tap selection could
    differ from the reference.  Checked on every 7th fp32 bit pattern for the principal points of all bench configs
    (an exhaustive run over all 2^32 patterns for 13 constants found 0 mismatches)."""
    from neuralrgbd_amd import camera
    from oracle import cpu_oracle as co
    cs = set()
    for cam in (camera.scannet_intrinsics(256, 192), camera.scannet_intrinsics(96, 64), camera.kitti_intrinsics(192, 64),
                camera.scannet_intrinsics(160, 120)):
        cs.add(float(cam["intrinsic_M"][0, 2])); cs.add(float(cam["intrinsic_M"][1, 2]))
    for c in sorted(cs):
        assert co.div_const_mismatches(c, stride=7) == 0, c


def test_stride2_conv_as_window_conv_on_space_to_depth_host_logic():
    """ops.conv_s2_weights (the re-indexing behind nrgbd_conv2d_taps_f32, taps = 4) in plain torch on the CPU: a stride-2 pad-1 3x3
    convolution equals the 2x2-window convolution {y-1, y} x {x-1, x} with the re-indexed weights on the space-to-depth input
    (channel (py*2+px)*C + c = x[2y+py, 2x+px, c]) — psm_submodule.py:90,120."""
    import torch
    import torch.nn.functional as F
    from neuralrgbd_amd import ops
    g = torch.Generator().manual_seed(0)
    for C, Cout, H, W in ((3, 32, 12, 20), (32, 64, 8, 10)):
        x = torch.randn(2, C, H, W, generator=g, dtype=torch.float64)
        w = torch.randn(Cout, C, 3, 3, generator=g, dtype=torch.float64)
        want = F.conv2d(x, w, stride=2, padding=1)
        w2 = ops.conv_s2_weights(w)
        cp = w2.shape[1]
        s2d = torch.zeros(2, cp, H // 2, W // 2, dtype=torch.float64)
        for py in (0, 1):
            for px in (0, 1):
                s2d[:, (py * 2 + px) * C:(py * 2 + px + 1) * C] = x[:, :, py::2, px::2]
        # window rows {y-1, y}, columns {x-1, x}: pad one pixel on the top / left only
        got = F.conv2d(F.pad(s2d, (1, 0, 1, 0)), w2)
        assert got.shape == want.shape
        assert (got - want).abs().max().item() < 1e-12


def test_graph_owners_pin_the_device_constants_they_captured():
    """homography's LRU of device constants (K, rays, d_candi) is bounded; a captured hipGraph reads those tensors by
    raw pointer and never refreshes the LRU, so the graph owner must hold them (cache_snapshot) — checked on the CPU side: the
    snapshot returns the very objects the cache holds, and they survive eviction."""
    import collections
    from neuralrgbd_amd import homography as H
    saved = H._const_cache
    try:
        H._const_cache = collections.OrderedDict()
        marker = object()
        H._cache_put(("d", b"x", "cpu"), marker)
        snap = H.cache_snapshot()
        assert snap == [marker]
        for i in range(H._CONST_CACHE_MAX + 5):
            H._cache_put(("d", b"y%d" % i, "cpu"), i)
        assert ("d", b"x", "cpu") not in H._const_cache and snap[0] is marker       # evicted from the LRU, alive in the snapshot
    finally:
        H._const_cache = saved
    import inspect
    from neuralrgbd_amd import streaming, train_step
    assert "cache_snapshot" in inspect.getsource(streaming.DepthStream._capture)
    assert "cache_snapshot" in inspect.getsource(train_step.TrainGraph.step)


def test_bench_refuses_a_traffic_measurement_taken_on_other_kernel_sources(tmp_path, monkeypatch):
    """VERDICT r2 (weak 9): roofline.traffic comes from a committed rocprofv3 --pmc measurement; bench.py reports it only while
    the sampling kernel's sources still hash to what the measurement recorded, and says why when it does not."""
    import json
    import bench
    h = bench.costvol_source_hash()
    assert len(h) == 16 and h == bench.costvol_source_hash()
    good = tmp_path / "good.json"
    good.write_text(json.dumps({"kernel_source_sha16": h, "B": {"traffic_bytes": 123}}))
    stale = tmp_path / "stale.json"
    stale.write_text(json.dumps({"kernel_source_sha16": "0" * 16, "B": {"traffic_bytes": 123}}))
    monkeypatch.setattr(bench, "TRAFFIC_FILE", str(good))
    assert bench.pmc_traffic("B")[0] == 123 and bench.pmc_traffic("S")[0] is None
    monkeypatch.setattr(bench, "TRAFFIC_FILE", str(stale))
    val, why = bench.pmc_traffic("B")
    assert val is None and "refused" in why
    monkeypatch.setattr(bench, "TRAFFIC_FILE", str(tmp_path / "missing.json"))
    assert bench.pmc_traffic("B")[0] is None


def test_live_traffic_measurement_parses_the_counter_files_and_fails_soft(tmp_path, monkeypatch):
    """bench.live_pmc_traffic: three rocprofv3 --pmc child passes (FETCH_SIZE, WRITE_SIZE, the SQ set) -> FETCH_SIZE x2 + WRITE_SIZE
    per dispatch of the sampling kernel + its mean SQ counters (-> roofline.valu_frac / lds_frac); a missing profiler, a failing
    child or an output without the kernel give (None, reason, None) and the caller falls back to the committed file."""
    import subprocess
    import bench
    calls = []

    def fake_run(cmd, **kw):
        calls.append(cmd)
        assert "--pmc" in cmd and "--no-live-traffic" in cmd and "--no-graph" in cmd      # the child cannot recurse
        counters, out = cmd[cmd.index("--pmc") + 1:cmd.index("--output-format")], cmd[cmd.index("-d") + 1]
        os.makedirs(os.path.join(out, "host", "1"), exist_ok=True)
        with open(os.path.join(out, "host", "1", "p_counter_collection.csv"), "w") as f:
            f.write("Kernel_Name,Counter_Name,Counter_Value\n")
            for counter in counters:
                for v in (100.0, 300.0):
                    f.write('"void nrgbd::costvol_quad<0, 3, false, 188, 3, false>(nrgbd::CostvolArgs)",%s,%f\n' % (counter, v))
                f.write('"other_kernel",%s,999999\n' % counter)
        return subprocess.CompletedProcess(cmd, 0)

    import shutil
    monkeypatch.setattr(shutil, "which", lambda name: sys.executable)       # any existing file stands in for the profiler
    monkeypatch.setattr(subprocess, "run", fake_run)
    val, note, sq = bench.live_pmc_traffic("B")
    assert val == int(2 * 200 * 1024 + 200 * 1024) and "measured in this run" in note and len(calls) == 3
    assert set(sq) == set(bench.SQ_PASS) and sq["SQ_ACTIVE_INST_VALU"] == 200.0
    fr = bench.sq_fractions({"GRBM_GUI_ACTIVE": 8e6, "SQ_ACTIVE_INST_VALU": 1.28e8, "SQ_LDS_IDX_ACTIVE": 6.4e7, "SQ_LDS_BANK_CONFLICT": 0.0,
                             "SQ_WAVE_CYCLES": 1e9, "SQ_WAIT_ANY": 2.5e8})
    assert abs(fr["valu_frac"] - 0.5) < 1e-9 and abs(fr["lds_frac"] - 0.25) < 1e-9 and abs(fr["waves_waiting_frac"] - 0.25) < 1e-9

    def failing(cmd, **kw):
        raise subprocess.CalledProcessError(1, cmd)
    monkeypatch.setattr(subprocess, "run", failing)
    val, note, sq = bench.live_pmc_traffic("B")
    assert val is None and "failed" in note and sq is None


def test_conv_embeddings_of_the_training_path_are_exact(monkeypatch):
    """autograd.conv2d_module / conv_transpose2d_module express the stride-2 3x3, the 1x1 (stride 1 / 2), the odd-width (67
    channel) and the ConvTranspose2d(k4, s2, p1) layers as ONE 3x3 stride-1 convolution on padded / space-to-depth tensors with
    an embedded weight (psm_submodule.py:90-139, Refine.py:51-77).  With F.conv2d standing in for the device kernel, forward,
    input gradient, weight gradient and bias gradient must equal the torch module's — the index maps are exact, only summation
    order differs."""
    import torch
    import torch.nn as nn
    import torch.nn.functional as F
    from neuralrgbd_amd import autograd as ag

    class FakeConv2dCL:
        calls = []

        @staticmethod
        def eligible(cin, cout, dil, need_dgrad=True, real=None):
            return cin % 16 == 0 and cout % 16 == 0

        @staticmethod
        def apply(x, w, dil, key=None, real=None):
            FakeConv2dCL.calls.append((tuple(x.shape), tuple(w.shape), dil))
            return F.conv2d(x, w, None, 1, dil, dil)
    monkeypatch.setattr(ag, "Conv2dCL", FakeConv2dCL)
    torch.manual_seed(0)
    cases = [
        (nn.Conv2d(3, 32, 3, 2, 1, bias=False), (2, 3, 16, 24), ag.conv2d_module),          # firstconv: image -> 12 -> 16 channels
        (nn.Conv2d(32, 64, 3, 2, 1, bias=False), (2, 32, 12, 8), ag.conv2d_module),         # layer2 entry
        (nn.Conv2d(32, 64, 1, 2, 0, bias=False), (2, 32, 12, 8), ag.conv2d_module),         # strided 1x1 shortcut
        (nn.Conv2d(128, 32, 1, 1, 0, bias=False), (1, 128, 5, 7), ag.conv2d_module),        # SPP branch
        (nn.Conv2d(67, 67, 3, 1, 1, bias=True), (1, 67, 10, 12), ag.conv2d_module),         # R-Net conv2 (67 -> 67, bias)
        (nn.Conv2d(67, 64, 3, 1, 1, bias=True), (1, 67, 10, 12), ag.conv2d_module),         # R-Net conv2_1
        (nn.Conv2d(64, 64, 3, 1, 2, 2, bias=False), (1, 64, 9, 11), ag.conv2d_module),      # dilated trunk layer: passes straight through
        (nn.ConvTranspose2d(128, 64, 4, 2, 1, bias=True), (1, 128, 6, 5), ag.conv_transpose2d_module),
        (nn.ConvTranspose2d(96, 64, 4, 2, 1, bias=True), (2, 96, 4, 7), ag.conv_transpose2d_module),
    ]
    for mod, shape, fn in cases:
        mod = mod.double()
        x = torch.randn(*shape, dtype=torch.float64, requires_grad=True)
        monkeypatch.setattr(torch, "float32", torch.float64)      # the dtype gate of the module functions (fp32 on the device)
        n0 = len(FakeConv2dCL.calls)
        y = fn(mod, x, _any_device=True)
        monkeypatch.undo(); monkeypatch.setattr(ag, "Conv2dCL", FakeConv2dCL)
        assert len(FakeConv2dCL.calls) == n0 + 1, "the layer must run as ONE stride-1 3x3 convolution"
        want = mod(x)
        assert y.shape == want.shape
        g = torch.randn_like(want)
        params = [mod.weight] + ([mod.bias] if mod.bias is not None else [])
        got_g = torch.autograd.grad(y, [x] + params, g)
        want_g = torch.autograd.grad(want, [x] + params, g)
        assert torch.allclose(y, want, rtol=1e-12, atol=1e-12), type(mod)
        for a, b in zip(got_g, want_g):
            assert torch.allclose(a, b, rtol=1e-11, atol=1e-11), (type(mod), a.shape)


def test_relu_unit_bounds_any_batchnorm_output():
    """ops.relu_unit (the power of two the clamped-FMA ReLU form of wino_dw.hip scales its operands by): 1 / unit exceeds the
    largest |BatchNorm(y)| under batch statistics for adversarial data — a single huge outlier, constant channels, tiny n — so the
    [0, 1] clamp can never saturate on the upper side."""
    import torch
    from neuralrgbd_amd import ops
    g = torch.Generator().manual_seed(5)
    for n in (2, 7, 4096, 100000):
        for trial in range(4):
            C = 6
            y = torch.randn(n, C, generator=g) * (10.0 ** trial)
            y[0, 0] = 1e7                      # one outlier dominates channel 0
            y[:, 1] = 3.25                     # a constant channel (variance 0: eps alone normalises)
            y[:, 2] = 0.0
            y[n // 2, 2] = -1e-3
            gamma = torch.randn(C, generator=g) * 3
            beta = torch.randn(C, generator=g) * 5
            z = torch.nn.functional.batch_norm(y, None, None, gamma, beta, training=True, eps=1e-5)
            unit = ops.relu_unit(gamma, beta, n)
            k = round(-__import__("math").log2(unit))
            assert unit == 2.0 ** -k and 0 <= k, unit              # an exact power of two in (0, 1]
            assert z.abs().max().item() < 1.0 / unit, (n, trial, z.abs().max().item(), unit)


def test_rnet_candidate_padding_embedding_is_exact():
    """DPVUpsampleNet._embedded (round 5: a net with D != 64 / 128 candidates runs zero-padded to 64 / 128 candidate channels on
    the hand-written kernels): the re-indexed weights, evaluated with plain torch convolutions on the padded channel layout
    [D real | Dp - D zero | image features], reproduce the module graph (Refine.py:79-107) — the padding contributes exact zeros,
    the -1e30 bias keeps it out of the final log-softmax.  Host logic only: no GPU, no kernel."""
    import torch.nn.functional as F
    from neuralrgbd_amd import nets, synth
    torch.manual_seed(0)
    for D in (16, 24, 100):
        net = nets.DPVUpsampleNet(64, 32, 3, D=D).double()
        net.load_state_dict({k: v.double() for k, v in synth.seeded_state_dict(net, 5).items()})
        for m in net.modules():
            if getattr(m, "bias", None) is not None:
                torch.nn.init.normal_(m.bias, 0, 0.1)
        Dr, Dp, C0, C1, C2 = net._widths()
        assert (Dr, Dp) == (D, 64 if D <= 64 else 128)
        h, w = 6, 8
        dpv = torch.softmax(torch.randn(1, D, h, w, dtype=torch.float64), 1)
        feats = [torch.randn(1, 64, h, w, dtype=torch.float64), torch.randn(1, 32, 2 * h, 2 * w, dtype=torch.float64),
                 torch.rand(1, 3, 4 * h, 4 * w, dtype=torch.float64)]
        with torch.no_grad():
            want = net(dpv, feats)                                   # the CPU module graph
            e = net._embedded()
            pad = lambda x, f: torch.cat((x, x.new_zeros(1, Dp - x.shape[1], *x.shape[2:]), f), 1)      # [D | zeros | features]
            cl = lambda x, k: F.leaky_relu(F.conv2d(x, e[k][0], e[k][1], 1, 1), 0.01)
            tl = lambda x, k: F.leaky_relu(F.conv_transpose2d(x, e[k][0], e[k][1], 2, 1), 0.01)
            x = cl(cl(pad(dpv, feats[0]), "conv0"), "conv0_1")
            assert bool((x[:, D:Dp] == 0).all())                     # the padding stays exactly zero through a layer
            x = tl(x, "trans_conv0")
            assert x.shape[1] == Dp and bool((x[:, D:] == 0).all())
            x = cl(cl(torch.cat((x, feats[1]), 1), "conv1"), "conv1_1")
            x = tl(x, "trans_conv1")
            x = cl(cl(torch.cat((x, feats[2]), 1), "conv2"), "conv2_1")
            z = F.conv2d(x, e["conv2_2"][0], e["conv2_2"][1], 1, 1)
            got = torch.log_softmax(z, 1)[:, :D]
        assert got.shape == want.shape
        assert (got - want).abs().max().item() < 1e-9, D
    assert nets.DPVUpsampleNet(64, 32, 3, D=200)._widths() is None and nets.DPVUpsampleNet(64, 32, 3, D=16, upsample_D=True)._widths() is None


def test_fused_adam_loads_a_reference_era_checkpoint():
    """the reference saves `optimizer.state_dict()` of a torch < 1.12 Adam (train_KVNet.py:347): every
    param_group holds only lr / betas / eps / weight_decay / amsgrad and `step` is a Python int.  torch's load REPLACES the groups,
    so the keys this optimizer's step() reads (`maximize`) must be filled from the defaults; an AMSGrad checkpoint is refused before
    any state is replaced.  (Host logic only: step() itself needs the GPU — tests/test_gpu_train.py steps after such a load.)"""
    from neuralrgbd_amd._lib import NrgbdError
    from neuralrgbd_amd.optim import FusedAdam
    ps = [torch.nn.Parameter(torch.randn(5, 3)), torch.nn.Parameter(torch.randn(7))]
    opt = FusedAdam(ps, lr=1e-3)
    old = {"state": {0: {"step": 11, "exp_avg": torch.full((5, 3), 0.5), "exp_avg_sq": torch.full((5, 3), 0.25)},
                     1: {"step": 11, "exp_avg": torch.zeros(7), "exp_avg_sq": torch.ones(7)}},
           "param_groups": [{"lr": 1e-5, "betas": (0.9, 0.999), "eps": 1e-8, "weight_decay": 0, "amsgrad": False, "params": [0, 1]}]}
    opt.load_state_dict(old)
    g = opt.param_groups[0]
    assert g["lr"] == 1e-5 and g["maximize"] is False and set(opt.defaults) <= set(g)
    assert opt.state[ps[0]]["step"] == 11 and float(opt.state[ps[0]]["exp_avg"][0, 0]) == 0.5
    import copy
    import pickle
    o2 = pickle.loads(pickle.dumps(opt))                     # __setstate__ fills the same defaults
    assert o2.param_groups[0]["maximize"] is False
    ams = copy.deepcopy(old)
    ams["param_groups"][0]["amsgrad"] = True
    fresh = FusedAdam(ps, lr=1e-3)
    with pytest.raises(NrgbdError):
        fresh.load_state_dict(ams)
    assert fresh.param_groups[0]["lr"] == 1e-3 and not fresh.state      # refused before anything was replaced


def test_rnet_route_plan_and_padded_widths_of_the_training_convolutions():
    """Host logic of autograd.Conv2dCL (round 6): which 3x3 layers of the training path run on wino_pc.hip's R-Net form and at which
    zero-padded widths — the R-Net's 67-wide layers at 80 (64 columns + 3 on conv_few.hip), its 96-wide ones as 64 + a 32-column slice,
    every trunk layer where it was."""
    from neuralrgbd_amd.autograd import Conv2dCL, _padded_widths
    plan = Conv2dCL._rnet_plan
    assert plan(80, 80, 1, 67) == (64, ("few", 3)) and plan(80, 64, 1, 64) == (64, None) and plan(64, 80, 1, 67) == (64, ("few", 3))
    assert plan(96, 96, 1) == (64, ("half", 32)) and plan(80, 80, 1, 72) == (64, ("half", 8)) and plan(144, 144, 1, 131) == (128, ("few", 3))
    assert plan(64, 64, 1) is None and plan(128, 128, 2) is None          # whole 64-column groups with Cin % 32 == 0: the plain form
    assert plan(80, 80, 2, 67) is None and plan(16, 80, 1, 67) is None and plan(80, 48, 1) is None and plan(80, 112, 1, 104) is None
    assert _padded_widths(67, 67, 1, True) == (80, 80) and _padded_widths(80, 64, 1, True, real=(67, 64)) == (80, 64)
    assert _padded_widths(96, 96, 1, True) == (96, 96) and _padded_widths(72, 72, 1, True) == (80, 80)
    for shape in ((64, 64, 1), (32, 32, 1), (128, 128, 2), (320, 128, 1), (64, 128, 1), (128, 128, 1)):
        assert _padded_widths(shape[0], shape[1], shape[2], True) == shape[:2], shape
    assert _padded_widths(12, 32, 1, False) == (16, 32)                    # the space-to-depth image of firstconv: no data gradient
    Conv2dCL.rnet_route = False
    try:
        assert plan(80, 80, 1, 67) is None and _padded_widths(67, 67, 1, True) == (96, 96)       # the direct 96-wide kernel (rounds 2-5)
    finally:
        Conv2dCL.rnet_route = True


def test_pack_cache_scopes():
    """autograd.pack_cache: streams are computed once per (layer key, what) inside a scope, never outside one, never without a key;
    nested scopes restore the outer one (TrainGraph captures under its own store while train() may hold another)."""
    from neuralrgbd_amd import autograd as ag
    calls = []

    def make(v):
        def fn():
            calls.append(v)
            return v
        return fn
    assert ag._cached(1, ("a", 0), make("x")) == "x" and ag._cached(1, ("a", 0), make("x")) == "x" and calls == ["x", "x"]     # no scope
    with ag.pack_cache() as outer:
        assert ag._cached(1, ("a", 0), make("y")) == "y" and ag._cached(1, ("a", 0), make("z")) == "y"
        assert ag._cached(None, ("a", 0), make("k")) == "k" and ag._cached(None, ("a", 0), make("k")) == "k"                    # no key
        assert ag._cached(1, ("a", 1), make("t")) == "t" and ag._cached(2, ("a", 0), make("u")) == "u"
        store = {}
        with ag.pack_cache(store):
            assert ag._cached(1, ("a", 0), make("w")) == "w" and store == {(1, ("a", 0)): "w"}
        assert ag._cached(1, ("a", 0), make("q")) == "y" and len(outer) == 3
    assert ag._PACK_CACHE is None
    assert calls == ["x", "x", "y", "k", "k", "t", "u", "w"]
