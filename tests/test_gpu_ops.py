"""HIP kernels (through the C-ABI of libnrgbd_hip.so) vs the CPU oracle and the golden vectors.

Tolerances (fp32, written here as the contract): cost / log-prob volumes within 1e-4 absolute
(BASELINE.json: "DPV floats within 1e-4"), arg-max / arg-min indices bit-exact except at near
ties of the oracle's own values (counted and bounded), PREDICT resample within 1e-4.
"""
import ctypes
import math

import numpy as np
import pytest
import torch

from conftest import near_tie_mismatches, report
from neuralrgbd_amd import camera, synth
from oracle import cpu_oracle as co
from oracle import gen_golden

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _ops():
    from neuralrgbd_amd import ops
    return ops


def _dev(a):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(DEV)


def _gpu_costvol(feat_ref, feat_src, KR, Kt, rays, d_candi, cx, cy, sigma, dist="L2", logp=False, align=False,
                 generation=None, want_cost=True):
    ops = _ops()
    V, C = feat_src.shape[:2]
    tex = ops.pack_nhwc(_dev(np.concatenate([feat_src, feat_ref[None]], 0)))
    cost, lp = ops.costvol(tex[V], tex[:V], _dev(KR), _dev(Kt), _dev(rays), _dev(d_candi), cx, cy, sigma, C,
                           dist=dist, align_corners=align, want_cost=want_cost, want_logp=logp, generation=generation)
    torch.cuda.synchronize()
    return (cost.cpu().numpy() if want_cost else None), (lp.cpu().numpy() if logp else None)


def test_library_is_the_hip_build():
    ops = _ops()
    assert "gfx950" in ops.version()
    assert torch.cuda.is_available() and "gfx95" in torch.cuda.get_device_properties(0).gcnArchName


def test_golden_costvol_and_logsoftmax(golden_ops):
    g = golden_ops
    o = gen_golden.OPS
    cam = camera.scannet_intrinsics(o["w"], o["h"])
    rays = cam["unit_ray_array_2D"].numpy()
    cx, cy = cam["intrinsic_M"][0, 2], cam["intrinsic_M"][1, 2]
    for dist, key in (("L2", "cost_l2"), ("L1", "cost_l1")):
        cost, lp = _gpu_costvol(g["feat_ref"], g["feat_src"], g["KR"], g["Kt"], rays, g["d_candi"], cx, cy,
                                float(g["sigma"]), dist=dist, logp=True)
        mx, _, mism = report("HIP costvol %s vs reference" % dist, -cost, -g[key])
        assert mx < 1e-4 and mism == 0
        if dist == "L2":
            mx, _, mism = report("HIP fused log-softmax vs reference", lp, g["bv"])
            assert mx < 1e-4 and mism == 0


@pytest.fixture(params=["lds", "gather"])
def costvol_generation(request):
    """The general-shape generations of the fused kernel must satisfy the same parity contract (the generation is an
    argument of nrgbd_costvol_fwd_gen; the quad generation, specific to the path's 64(+3)-channel texel, has its own
    tests below)."""
    return request.param


@pytest.mark.parametrize("h,w,D,V,C,seed", [
    (24, 40, 16, 4, 11, 1),      # appendix-C shape
    (17, 23, 5, 1, 3, 2),        # ragged: nothing divides the tile sizes, single view, one 16-B word
    (33, 70, 64, 5, 67, 3),      # the real channel count, 5 source views
    (64, 96, 64, 4, 67, 4),      # reference-native ScanNet grid (config S)
    (8, 8, 2, 2, 4, 5),          # tiny
    (40, 56, 130, 2, 9, 6),      # D > 128
])
def test_costvol_vs_oracle(h, w, D, V, C, seed, costvol_generation):
    cam = camera.scannet_intrinsics(w, h)
    rng = np.random.RandomState(seed)
    feat_ref = rng.standard_normal((C, h, w)).astype(np.float32)
    feat_src = rng.standard_normal((V, C, h, w)).astype(np.float32)
    poses = synth.random_poses(rng, V)
    KR, Kt = co.homography_terms(cam["intrinsic_M_cuda"].numpy(), poses[:, :3, :3], poses[:, :3, 3])
    d_candi = np.linspace(0.1, 5, D)
    rays = cam["unit_ray_array_2D"].numpy()
    cx, cy = cam["intrinsic_M"][0, 2], cam["intrinsic_M"][1, 2]
    want = co.costvol(feat_ref, feat_src, KR, Kt, rays, d_candi, cx, cy, 10.0)
    want_lp = co.logsoftmax_d(want, scale=-1.0)
    cost, lp = _gpu_costvol(feat_ref, feat_src, KR, Kt, rays, d_candi, cx, cy, 10.0, logp=True,
                            generation=costvol_generation)
    mx, _, _ = report("HIP costvol %dx%dx%d V%d C%d" % (h, w, D, V, C), -cost, -want)
    assert mx < 1e-4
    mx, mean, _ = report("HIP BV_cur", lp, want_lp)
    assert mx < 1e-4 and mean < 1e-5
    assert near_tie_mismatches(lp, want_lp, tol=1e-4) == 0


def test_costvol_out_of_view_and_align_corners(costvol_generation):
    """Large motions push most taps outside the source image (zeros padding); legacy align_corners path."""
    h, w, D, V, C = 20, 28, 8, 3, 6
    cam = camera.scannet_intrinsics(w, h)
    rng = np.random.RandomState(9)
    feat_ref = rng.standard_normal((C, h, w)).astype(np.float32)
    feat_src = rng.standard_normal((V, C, h, w)).astype(np.float32)
    poses = synth.random_poses(rng, V, rot_sigma=0.4, trans_sigma=1.0)
    KR, Kt = co.homography_terms(cam["intrinsic_M_cuda"].numpy(), poses[:, :3, :3], poses[:, :3, 3])
    d_candi = np.linspace(0.1, 5, D)
    rays = cam["unit_ray_array_2D"].numpy()
    cx, cy = cam["intrinsic_M"][0, 2], cam["intrinsic_M"][1, 2]
    for align in (False, True):
        want = co.costvol(feat_ref, feat_src, KR, Kt, rays, d_candi, cx, cy, 1.0, align_corners=align)
        got, _ = _gpu_costvol(feat_ref, feat_src, KR, Kt, rays, d_candi, cx, cy, 1.0, align=align,
                              generation=costvol_generation)
        assert np.abs(got - want).max() < 1e-4


def test_costvol_extreme_zoom_uses_gather_fallback():
    """Forward motion comparable to the nearest depth: the tile footprint of the nearest candidates
    exceeds the LDS patch (and for some tiles the plane crosses the source camera), which exercises the
    per-candidate split and the direct-gather fallback of the LDS generation."""
    h, w, D, V, C = 48, 64, 16, 2, 67
    cam = camera.scannet_intrinsics(w, h)
    rng = np.random.RandomState(13)
    feat_ref = rng.standard_normal((C, h, w)).astype(np.float32)
    feat_src = rng.standard_normal((V, C, h, w)).astype(np.float32)
    poses = synth.random_poses(rng, V)
    poses[0, :3, 3] = (0.01, -0.02, -0.085)  # source camera 8.5 cm in front: 6.7x zoom at d = 0.1
    poses[1, :3, 3] = (0.02, 0.01, 0.12)     # ... and 12 cm behind
    KR, Kt = co.homography_terms(cam["intrinsic_M_cuda"].numpy(), poses[:, :3, :3], poses[:, :3, 3])
    d_candi = np.linspace(0.1, 5, D)
    rays = cam["unit_ray_array_2D"].numpy()
    cx, cy = cam["intrinsic_M"][0, 2], cam["intrinsic_M"][1, 2]
    want = co.costvol(feat_ref, feat_src, KR, Kt, rays, d_candi, cx, cy, 10.0)
    got, _ = _gpu_costvol(feat_ref, feat_src, KR, Kt, rays, d_candi, cx, cy, 10.0)
    mx, _, _ = report("HIP costvol extreme zoom", -got, -want)
    assert mx < 1e-4


def test_identity_pose_reproduces_reference_features():
    """R = I, t = 0: every candidate samples the pixel itself, so the cost is ~0 everywhere."""
    h, w, D, C = 16, 24, 4, 8
    cam = camera.scannet_intrinsics(w, h)
    rng = np.random.RandomState(10)
    feat = rng.standard_normal((C, h, w)).astype(np.float32)
    poses = np.eye(4, dtype=np.float32)[None]
    KR, Kt = co.homography_terms(cam["intrinsic_M_cuda"].numpy(), poses[:, :3, :3], poses[:, :3, 3])
    got, _ = _gpu_costvol(feat, feat[None], KR, Kt, cam["unit_ray_array_2D"].numpy(), np.linspace(.1, 5, D),
                          cam["intrinsic_M"][0, 2], cam["intrinsic_M"][1, 2], 1.0)
    assert got.max() < 1e-6


def test_pack_nhwc_with_pooled_rgb():
    ops = _ops()
    rng = np.random.RandomState(11)
    N, Cf, h, w, pool = 3, 6, 10, 70, 4
    feat = rng.standard_normal((N, Cf, h, w)).astype(np.float32)
    rgb = rng.standard_normal((N, 3, h * pool, w * pool)).astype(np.float32)
    tex = ops.pack_nhwc(_dev(feat), _dev(rgb)).cpu().numpy()
    assert tex.shape == (N, h, w, 12)
    assert np.array_equal(tex[..., :Cf], feat.transpose(0, 2, 3, 1))
    assert np.abs(tex[..., Cf:Cf + 3] - co.avgpool(rgb, pool).transpose(0, 2, 3, 1)).max() < 1e-6
    assert np.all(tex[..., Cf + 3:] == 0)


def test_warp_volume_golden_and_assembly(golden_ops):
    ops = _ops()
    g = golden_ops
    o = gen_golden.OPS
    cam = camera.scannet_intrinsics(o["w"], o["h"])
    V, D, h, w = o["V"], o["D"], o["h"], o["w"]
    rgb = _dev(g["rgb"])
    bv_cur, bv_pred = _dev(g["bv"]), _dev(g["pred"])
    ref = _dev(g["rgb"][0])
    vol = ops.warp_volume(rgb, (3 * h * w, h * w, w, 1), ref, (h * w, w, 1), _dev(g["KR"]), _dev(g["Kt"]),
                          _dev(cam["unit_ray_array_2D"].numpy()), _dev(g["d_candi"]),
                          cam["intrinsic_M"][0, 2], cam["intrinsic_M"][1, 2], V, 3, h, w,
                          bv_cur=bv_cur, bv_pred=bv_pred).cpu().numpy()
    assert vol.shape == (3 * V + 4, D, h, w)
    assert np.abs(vol[:3 * V].reshape(V, 3, D, h, w) - g["warped"]).max() < 1e-5
    assert np.array_equal(vol[3 * V:3 * V + 3], np.broadcast_to(g["rgb"][0][:, None], (3, D, h, w)))
    assert np.array_equal(vol[-1], g["bv"] - g["pred"])


def test_warp_volume_channels_last_matches_planar():
    """The K-Net assembly in the conv3d layout ([D,h,w,16], fast path on the RGB word of the texel tensor)
    must equal the planar torch layout."""
    ops = _ops()
    rng = np.random.RandomState(21)
    V, h, w, D = 4, 20, 36, 8
    cam = camera.scannet_intrinsics(w, h)
    tex = _dev(rng.standard_normal((V + 1, h, w, 68)))
    poses = synth.random_poses(rng, V)
    KR, Kt = co.homography_terms(cam["intrinsic_M_cuda"].numpy(), poses[:, :3, :3], poses[:, :3, 3])
    bv, bp = _dev(rng.standard_normal((D, h, w))), _dev(rng.standard_normal((D, h, w)))
    args = (tex[:V, :, :, 64:], (h * w * 68, 1, w * 68, 68), tex[V, :, :, 64:], (1, w * 68, 68), _dev(KR), _dev(Kt),
            _dev(cam["unit_ray_array_2D"].numpy()), _dev(np.linspace(.1, 5, D)), cam["intrinsic_M"][0, 2],
            cam["intrinsic_M"][1, 2], V, 3, h, w)
    planar = ops.warp_volume(*args, bv_cur=bv, bv_pred=bp)
    cl = ops.warp_volume(*args, bv_cur=bv, bv_pred=bp, channels_last=True)
    assert cl.shape == (D, h, w, 16)
    assert torch.equal(cl.permute(3, 0, 1, 2), planar)


def test_dpv_resample_golden_and_oracle(golden_ops):
    ops = _ops()
    g = golden_ops
    o = gen_golden.OPS
    cam = camera.scannet_intrinsics(o["w"], o["h"])
    from neuralrgbd_amd import homography as H
    z_half, z_rad = H.z_range(g["d_candi"])
    tan_hh, tan_hv = math.tan(math.radians(cam["hfov"]) * .5), math.tan(math.radians(cam["vfov"]) * .5)
    got = ops.dpv_resample(_dev(g["dpv"]), _dev(g["T"]), _dev(cam["unit_ray_array_2D"].numpy()), _dev(g["d_candi"]),
                           tan_hh, tan_hv, z_half, z_rad, float(g["pad"])).cpu().numpy()
    mx, mean, mism = report("HIP PREDICT vs reference", got, g["pred"])
    assert mx < 1e-4 and mism == 0
    # a larger motion: many points leave the frustum (border clamp), others hit the padded faces
    rng = np.random.RandomState(12)
    T = np.linalg.inv(synth.random_pose(rng, 0.2, 0.5)).astype(np.float32)
    want = co.dpv_resample(g["dpv"], T, cam["unit_ray_array_2D"].numpy(), g["d_candi"], tan_hh, tan_hv, float(g["pad"]))
    got = ops.dpv_resample(_dev(g["dpv"]), _dev(T), _dev(cam["unit_ray_array_2D"].numpy()), _dev(g["d_candi"]),
                           tan_hh, tan_hv, z_half, z_rad, float(g["pad"])).cpu().numpy()
    assert np.abs(got - want).max() < 1e-4


@pytest.mark.parametrize("D,n", [(64, 1000), (16, 37), (128, 513), (200, 70)])
def test_logsoftmax_update_and_depth_regress(D, n):
    ops = _ops()
    rng = np.random.RandomState(D + n)
    a = (rng.standard_normal((D, n)) * 5).astype(np.float32)
    b = np.log(np.abs(rng.standard_normal((D, n))) + 1e-3).astype(np.float32)
    got = ops.logsoftmax_d(_dev(a), _dev(b)).cpu().numpy()
    want = co.logsoftmax_d(a, b)
    assert np.abs(got - want).max() < 2e-5
    assert np.abs(np.exp(got.astype(np.float64)).sum(0) - 1).max() < 1e-5
    d = np.linspace(.1, 5, D)
    depth, conf = ops.depth_regress(_dev(want), _dev(d))
    wd, wc = co.depth_regress(want, d)
    assert np.abs(depth.cpu().numpy() - wd).max() < 1e-5 and np.array_equal(conf.cpu().numpy(), wc)


def test_argument_errors_and_cpu_tensors_fail_loudly():
    from neuralrgbd_amd import _lib
    ops = _ops()
    with pytest.raises(_lib.NrgbdError):
        ops.logsoftmax_d(torch.zeros(4, 4))  # CPU tensor: no fallback
    lib = _lib.load()
    assert lib.nrgbd_logsoftmax_d(None, None, ctypes.c_float(1), None, 4, 4, None) == -1
    x = torch.zeros(8, 8, device=DEV)
    assert lib.nrgbd_logsoftmax_d(ctypes.c_void_p(x.data_ptr()), None, ctypes.c_float(1),
                                  ctypes.c_void_p(x.data_ptr()), 0, 8, None) == -2


def test_limits_and_degenerate_sizes():
    """Maximum candidate / view counts of the C-ABI (NRGBD_MAX_D = 256, NRGBD_MAX_V = 16), D = 1, a 1-pixel-wide grid,
    and the error codes just beyond the limits."""
    from neuralrgbd_amd import _lib
    ops = _ops()
    rng = np.random.RandomState(31)
    for (h, w, D, V, C) in ((6, 10, 256, 2, 5), (5, 7, 3, 16, 4), (9, 1, 1, 1, 3), (1, 9, 2, 1, 67)):
        cam = camera.scannet_intrinsics(max(w, 2), max(h, 2)) if min(h, w) < 2 else camera.scannet_intrinsics(w, h)
        cam = camera.make_cam_intrinsics(cam["hfov"], cam["vfov"], w, h)
        feat_ref = rng.standard_normal((C, h, w)).astype(np.float32)
        feat_src = rng.standard_normal((V, C, h, w)).astype(np.float32)
        poses = synth.random_poses(rng, V)
        KR, Kt = co.homography_terms(cam["intrinsic_M_cuda"].numpy(), poses[:, :3, :3], poses[:, :3, 3])
        d = np.linspace(0.2, 6, D) if D > 1 else np.array([1.5])
        rays = cam["unit_ray_array_2D"].numpy()
        cx, cy = cam["intrinsic_M"][0, 2], cam["intrinsic_M"][1, 2]
        want = co.costvol(feat_ref, feat_src, KR, Kt, rays, d, cx, cy, 2.0)
        got, lp = _gpu_costvol(feat_ref, feat_src, KR, Kt, rays, d, cx, cy, 2.0, logp=True)
        assert np.abs(got - want).max() < 1e-4 * max(1.0, np.abs(want).max()), (h, w, D, V, C)
        assert np.abs(lp - co.logsoftmax_d(want, scale=-1.0)).max() < 2e-4
    lib = _lib.load()
    x = torch.zeros(64, device=DEV)
    p = ctypes.c_void_p(x.data_ptr())
    f = ctypes.c_float
    assert lib.nrgbd_costvol_fwd(p, p, p, p, p, p, f(1), f(1), f(1), 0, 0, p, None, 1, 4, 4, 257, 2, 2, None) == -2   # D > MAX_D
    assert lib.nrgbd_costvol_fwd(p, p, p, p, p, p, f(1), f(1), f(1), 0, 0, p, None, 17, 4, 4, 2, 2, 2, None) == -2   # V > MAX_V
    assert lib.nrgbd_costvol_fwd(p, p, p, p, p, p, f(1), f(1), f(1), 0, 0, p, None, 1, 4, 6, 2, 2, 2, None) == -3    # Cp % 4
    assert lib.nrgbd_costvol_fwd(p, p, p, p, p, p, f(1), f(1), f(1), 7, 0, p, None, 1, 4, 4, 2, 2, 2, None) == -4    # metric
    assert lib.nrgbd_costvol_fwd(p, p, p, p, p, p, f(1), f(1), f(1), 0, 0, None, None, 1, 4, 4, 2, 2, 2, None) == -1  # no output


def test_operator_surface_matches_reference_signatures(golden_ops):
    """neuralrgbd_amd.homography called exactly like warping.homography (tensors on the GPU) reproduces the golden
    outputs of est_swp_volume_v4 / warp_img_feats_v3 / resample_vol_cuda — the Level-2 integration of INTEGRATION.md."""
    from neuralrgbd_amd import homography as Hm
    g = golden_ops
    o = gen_golden.OPS
    cam = camera.scannet_intrinsics(o["w"], o["h"])
    poses = torch.from_numpy(g["poses"]).to(DEV)
    R, t = poses[:, :3, :3].contiguous(), poses[:, :3, 3].contiguous()
    cost = Hm.est_swp_volume_v4(_dev(g["feat_ref"])[None], _dev(g["feat_src"])[None], g["d_candi"], R, t, cam, float(g["sigma"]))
    assert cost.shape == (1, o["D"], o["h"], o["w"]) and np.abs(cost[0].cpu().numpy() - g["cost_l2"]).max() < 1e-4
    V = o["V"]
    warped = Hm.warp_img_feats_v3([_dev(g["rgb"][v:v + 1]) for v in range(V)], g["d_candi"], [R[v] for v in range(V)],
                                  [t[v] for v in range(V)], cam)
    assert len(warped) == V and tuple(warped[0].shape) == (3, o["D"], o["h"], o["w"])
    assert np.abs(torch.stack(warped).cpu().numpy() - g["warped"]).max() < 1e-4
    K = cam["intrinsic_M_cuda"].to(DEV)[None]
    rays = cam["unit_ray_array_2D"].to(DEV)[None]
    warped2 = Hm.warp_img_feats_mgpu([_dev(g["rgb"][v:v + 1]) for v in range(V)], g["d_candi"], [R[v] for v in range(V)],
                                     [t[v] for v in range(V)], K, rays)
    assert np.abs(torch.stack(warped2).cpu().numpy() - g["warped"]).max() < 1e-4
    pred = Hm.resample_vol_cuda(_dev(g["dpv"])[None], _dev(g["T"]), cam_intrinsic=cam, d_candi=g["d_candi"],
                                padding_value=float(g["pad"])).clamp(max=0, min=-1000.)
    assert np.array_equal(pred.cpu().numpy(), g["pred"])
    with pytest.raises(Exception):
        Hm.est_swp_volume_v4(_dev(g["feat_ref"])[None], _dev(g["feat_src"])[None], g["d_candi"], R, t, cam, 1.0, feat_dist="cosine")


def test_homography_terms_match_oracle_bitwise():
    """nrgbd_homography_terms == the C oracle (the reference's CPU summation order, pinned in code) bit for bit, on strided
    views of a pose tensor.  (torch's own CPU matmul on THIS host may order the K=3 sums differently: not the yardstick.)"""
    from neuralrgbd_amd import ops
    rng = np.random.RandomState(3)
    cam = camera.scannet_intrinsics(96, 64)
    K = cam["intrinsic_M_cuda"]
    poses = torch.from_numpy(synth.random_poses(rng, 5, rot_sigma=0.3, trans_sigma=0.5))
    KR, Kt = ops.homography_terms(K.cuda(), poses.cuda()[:, :3, :3], poses.cuda()[:, :3, 3])
    want_KR, want_Kt = co.homography_terms(K.numpy(), poses[:, :3, :3].numpy(), poses[:, :3, 3].numpy())
    assert np.array_equal(KR.cpu().numpy().reshape(5, 9), want_KR) and np.array_equal(Kt.cpu().numpy(), want_Kt)


# ----------------------------------------------------------------------------- generation 3 (quad) of the fused kernel
@pytest.mark.parametrize("h,w,D,V,C,dist,rot,trans,seed", [
    (64, 96, 64, 4, 67, "L2", 0.02, 0.05, 11),    # config S grid: candidate chunks + separate log-softmax
    (192, 256, 64, 4, 67, "L2", 0.02, 0.05, 12),  # config B grid: one workgroup per tile, fused log-softmax
    (33, 70, 64, 5, 67, "L2", 0.02, 0.05, 13),    # ragged tiles, 5 views
    (17, 23, 5, 1, 67, "L1", 0.02, 0.05, 14),     # tiny, single view, L1 metric, D < 8
    (40, 56, 130, 2, 67, "L2", 0.02, 0.05, 15),   # D > 128
    (120, 160, 128, 4, 67, "L2", 0.02, 0.05, 16), # config H grid, D = 128
    (24, 40, 16, 8, 64, "L2", 0.02, 0.05, 17),    # no RGB word (C = Cp = 64), 8 views
    (24, 40, 16, 3, 65, "L1", 0.02, 0.05, 18),    # one valid channel in the RGB word
    (48, 64, 32, 3, 67, "L2", 0.4, 1.0, 19),      # large motions: most taps out of view, planes crossing the camera
    (48, 64, 32, 4, 67, "L2", 0.0, 0.3, 20),      # pure translation 0.3 m: strong zoom on the nearest planes
    (192, 256, 48, 2, 67, "L2", 0.02, 0.05, 21),  # one workgroup per tile, two accumulator passes with a PARTIAL second one (32 + 16 candidates)
    (192, 256, 33, 1, 67, "L1", 0.02, 0.05, 22),  # ... a second pass of ONE candidate, single view
])
def test_costvol_quad_vs_oracle(h, w, D, V, C, dist, rot, trans, seed, gen="quad"):
    """Generation 3 of the fused kernel (4 lanes per (pixel, candidate); what the path runs) against the C oracle, against the
    automatic choice (must be the same kernel: identical bits) and against generation 2 (an independent decomposition)."""
    cam = camera.scannet_intrinsics(w, h)
    rng = np.random.RandomState(seed)
    feat_ref = rng.standard_normal((C, h, w)).astype(np.float32)
    feat_src = rng.standard_normal((V, C, h, w)).astype(np.float32)
    poses = synth.random_poses(rng, V, rot_sigma=rot, trans_sigma=trans)
    KR, Kt = co.homography_terms(cam["intrinsic_M_cuda"].numpy(), poses[:, :3, :3], poses[:, :3, 3])
    d_candi = np.linspace(0.1, 5, D)
    rays = cam["unit_ray_array_2D"].numpy()
    cx, cy = cam["intrinsic_M"][0, 2], cam["intrinsic_M"][1, 2]
    co.set_threads(32)
    want = co.costvol(feat_ref, feat_src, KR, Kt, rays, d_candi, cx, cy, 10.0, dist=dist)
    want_lp = co.logsoftmax_d(want, scale=-1.0)
    cost, lp = _gpu_costvol(feat_ref, feat_src, KR, Kt, rays, d_candi, cx, cy, 10.0, dist=dist, logp=True, generation=gen)
    _, lp_only = _gpu_costvol(feat_ref, feat_src, KR, Kt, rays, d_candi, cx, cy, 10.0, dist=dist, logp=True,
                              generation=gen, want_cost=False)        # generation 3 parks the raw costs in out_logp
    mx, _, _ = report("quad costvol %dx%dx%d V%d C%d %s" % (h, w, D, V, C, dist), -cost, -want)
    assert mx < 1e-5 * max(10.0, float(np.abs(want).max()))
    mx, mean, _ = report("quad BV_cur", lp, want_lp)
    assert mx < 1e-4 and mean < 1e-5
    assert near_tie_mismatches(lp, want_lp, tol=1e-4) == 0
    assert np.array_equal(lp, lp_only)
    c0, l0 = _gpu_costvol(feat_ref, feat_src, KR, Kt, rays, d_candi, cx, cy, 10.0, dist=dist, logp=True)   # automatic
    assert np.array_equal(c0, cost) and np.array_equal(l0, lp)
    if C == 67:   # against generation 2 on the same inputs (independent decomposition of the same arithmetic)
        c2, _ = _gpu_costvol(feat_ref, feat_src, KR, Kt, rays, d_candi, cx, cy, 10.0, dist=dist, generation="lds")
        assert np.abs(c2 - cost).max() < 1e-5 * max(10.0, float(np.abs(want).max()))


@pytest.mark.parametrize("kind", ["unit_norm", "barrel", "wavy"])
def test_costvol_quad_non_affine_ray_tables_vs_oracle(kind):
    """the C ABI takes ANY ray table, the staged patches of generation 3 are predicted from the images of a tile's four
    corner pixels — exact only for the pinhole table (rays affine in (x, y), z = 1).  With unit-norm rays, a barrel-distorted table
    and a deliberately wavy one (a tile's interior bulges several texels out of its corners' hull) the kernel must still equal
    the C oracle: a group whose tap leaves the predicted box is re-evaluated from global memory (costvol_quad.hip `escaped`)."""
    h, w, D, V, C = 96, 128, 64, 3, 67
    cam = camera.scannet_intrinsics(w, h)
    rng = np.random.RandomState(77)
    feat_ref = rng.standard_normal((C, h, w)).astype(np.float32)
    feat_src = rng.standard_normal((V, C, h, w)).astype(np.float32)
    poses = synth.random_poses(rng, V)
    KR, Kt = co.homography_terms(cam["intrinsic_M_cuda"].numpy(), poses[:, :3, :3], poses[:, :3, 3])
    d_candi = np.linspace(0.1, 5, D)
    rays = cam["unit_ray_array_2D"].numpy().astype(np.float64).reshape(3, h, w)
    if kind == "unit_norm":
        rays = rays / np.linalg.norm(rays, axis=0, keepdims=True)
    elif kind == "barrel":
        r2 = rays[0] ** 2 + rays[1] ** 2
        rays = np.stack([rays[0] * (1 + 0.25 * r2), rays[1] * (1 + 0.25 * r2), rays[2]])
    else:   # a 6-pixel-period ripple of ~2.5 texels amplitude: interior pixels of an 8x8 tile leave the corners' hull by whole texels
        ys, xs = np.meshgrid(np.arange(h), np.arange(w), indexing="ij")
        step = rays[0, 0, 1] - rays[0, 0, 0]
        rays = np.stack([rays[0] + 2.5 * step * np.sin(xs * 1.1 + ys * 0.7), rays[1] + 2.5 * step * np.cos(xs * 0.9 - ys * 1.3), rays[2]])
    rays = rays.reshape(3, h * w).astype(np.float32)
    cx, cy = cam["intrinsic_M"][0, 2], cam["intrinsic_M"][1, 2]
    co.set_threads(32)
    want = co.costvol(feat_ref, feat_src, KR, Kt, rays, d_candi, cx, cy, 10.0)
    got, lp = _gpu_costvol(feat_ref, feat_src, KR, Kt, rays, d_candi, cx, cy, 10.0, logp=True, generation="quad")
    auto, _ = _gpu_costvol(feat_ref, feat_src, KR, Kt, rays, d_candi, cx, cy, 10.0)
    gather, _ = _gpu_costvol(feat_ref, feat_src, KR, Kt, rays, d_candi, cx, cy, 10.0, generation="gather")
    mx, _, _ = report("quad costvol, %s rays" % kind, -got, -want)
    assert np.isfinite(got).all() and mx < 1e-5 * max(10.0, float(np.abs(want).max()))
    assert np.array_equal(auto, got)
    assert np.abs(gather - got).max() < 1e-5 * max(10.0, float(np.abs(want).max()))
    assert near_tie_mismatches(lp, co.logsoftmax_d(want, scale=-1.0), tol=1e-4) == 0


def test_costvol_quad_align_corners_and_determinism():
    h, w, D, V, C = 40, 72, 24, 4, 67
    cam = camera.scannet_intrinsics(w, h)
    rng = np.random.RandomState(31)
    feat_ref = rng.standard_normal((C, h, w)).astype(np.float32)
    feat_src = rng.standard_normal((V, C, h, w)).astype(np.float32)
    poses = synth.random_poses(rng, V)
    KR, Kt = co.homography_terms(cam["intrinsic_M_cuda"].numpy(), poses[:, :3, :3], poses[:, :3, 3])
    d_candi = np.linspace(0.1, 5, D)
    rays = cam["unit_ray_array_2D"].numpy()
    cx, cy = cam["intrinsic_M"][0, 2], cam["intrinsic_M"][1, 2]
    for align in (False, True):
        want = co.costvol(feat_ref, feat_src, KR, Kt, rays, d_candi, cx, cy, 3.0, align_corners=align)
        a, _ = _gpu_costvol(feat_ref, feat_src, KR, Kt, rays, d_candi, cx, cy, 3.0, align=align, generation="quad")
        b, _ = _gpu_costvol(feat_ref, feat_src, KR, Kt, rays, d_candi, cx, cy, 3.0, align=align, generation="quad")
        assert np.array_equal(a, b)   # no races, no atomics: bitwise reproducible
        assert np.abs(a - want).max() < 1e-5 * max(10.0, float(want.max()))


def test_costvol_generation_errors():
    """An explicit generation that does not support the shape is refused (NRGBD_E_SHAPE), never substituted."""
    from neuralrgbd_amd import _lib
    h, w, D, V, C = 16, 16, 4, 2, 11
    cam = camera.scannet_intrinsics(w, h)
    rng = np.random.RandomState(1)
    feat_ref = rng.standard_normal((C, h, w)).astype(np.float32)
    feat_src = rng.standard_normal((V, C, h, w)).astype(np.float32)
    poses = synth.random_poses(rng, V)
    KR, Kt = co.homography_terms(cam["intrinsic_M_cuda"].numpy(), poses[:, :3, :3], poses[:, :3, 3])
    with pytest.raises(_lib.NrgbdError):
        _gpu_costvol(feat_ref, feat_src, KR, Kt, cam["unit_ray_array_2D"].numpy(), np.linspace(.1, 5, D), 8.0, 8.0, 10.0,
                     generation="quad")


# ----------------------------------------------------------------------------- LBA depth warp (f4) and export epilogue (f2)
def test_lba_depth_warp_vs_reference_golden():
    """nrgbd_warp_depth_fwd/_bwd through the reference-named operator (autograd) vs the reference's output and torch
    autograd's gradients w.r.t. the poses (tests/golden/lba_small.npz), incl. the masked-L1 loss of opt_pose_numerical.py."""
    import os
    from conftest import GOLDEN
    from neuralrgbd_amd import homography as Hm
    o = gen_golden.LBA
    g = dict(np.load(os.path.join(GOLDEN, "lba_small.npz")))
    src, ref_img, dmap, poses, G = (torch.from_numpy(x).to(DEV) for x in gen_golden.lba_inputs())
    cam = camera.scannet_intrinsics(o["W"], o["H"])
    Rs = poses[:, :3, :3].clone().requires_grad_(True)
    ts = poses[:, :3, 3].clone().requires_grad_(True)
    out = Hm.back_warp_th_Rt_msrc(src, dmap, Rs, ts, cam)
    err = (out.detach().cpu().numpy() - g["warped"])
    print("[parity] LBA warp vs reference max|d|=%.2e" % np.abs(err).max())
    assert np.abs(err).max() < 2e-5
    (out * G).sum().backward()
    eR = np.abs(Rs.grad.cpu().numpy() - g["g_R"]).max() / np.abs(g["g_R"]).max()
    et = np.abs(ts.grad.cpu().numpy() - g["g_t"]).max() / np.abs(g["g_t"]).max()
    print("[parity] LBA warp gradients vs torch autograd: rel dR %.2e dt %.2e" % (eR, et))
    assert eR < 1e-4 and et < 1e-4
    Rs.grad = None; ts.grad = None
    out2 = Hm.back_warp_th_Rt_msrc(src, dmap, Rs, ts, cam)
    mask = 1.0 - (out2 == 0).type_as(out2)
    loss = torch.nn.L1Loss()(out2 * mask.detach(), ref_img * mask.detach())
    loss.backward()
    assert abs(float(loss) - float(g["loss"])) < 1e-5
    assert np.abs(Rs.grad.cpu().numpy() - g["g_R_loss"]).max() < 1e-3 * np.abs(g["g_R_loss"]).max()
    assert np.abs(ts.grad.cpu().numpy() - g["g_t_loss"]).max() < 1e-3 * np.abs(g["g_t_loss"]).max()
    single = Hm.back_warp_th_Rt(src[:1], dmap, poses[0, :3, :3], poses[0, :3, 3], cam)
    assert np.abs(single.cpu().numpy() - g["single"]).max() < 2e-5


def test_lba_depth_warp_vs_oracle_full_res_and_deterministic():
    """A full-resolution frame (480x640, 4 sources) against the C oracle; the gradient reduction is bitwise reproducible."""
    from neuralrgbd_amd import ops
    N, C, H, W = 4, 3, 480, 640
    rng = np.random.RandomState(8)
    cam = camera.scannet_intrinsics(W, H)
    src = rng.standard_normal((N, C, H, W)).astype(np.float32)
    dmap = (0.4 + 4.0 * rng.rand(H, W)).astype(np.float32)
    poses = synth.random_poses(rng, N, rot_sigma=0.02, trans_sigma=0.05)
    G = rng.standard_normal(src.shape).astype(np.float32)
    K, rays = cam["intrinsic_M_cuda"].numpy(), cam["unit_ray_array_2D"].numpy()
    R, t = np.ascontiguousarray(poses[:, :3, :3]), np.ascontiguousarray(poses[:, :3, 3])
    want = co.warp_depth_fwd(src, dmap, K, R, t, rays)
    wR, wt = co.warp_depth_bwd(src, dmap, K, R, t, rays, G)
    a = [_dev(x) for x in (src, dmap, K, R, t, rays)]
    got = ops.warp_depth_fwd(*a)
    assert np.abs(got.cpu().numpy() - want).max() < 2e-5
    gR, gt = ops.warp_depth_bwd(*a, _dev(G))
    gR2, gt2 = ops.warp_depth_bwd(*a, _dev(G))
    assert torch.equal(gR, gR2) and torch.equal(gt, gt2)
    assert np.abs(gR.cpu().numpy() - wR).max() < 1e-4 * np.abs(wR).max()
    assert np.abs(gt.cpu().numpy() - wt).max() < 1e-4 * np.abs(wt).max()


def test_export_epilogue_vs_reference_files_and_oracle(tmp_path):
    """nrgbd_export_depth_u16 vs the .pgm images the reference's export_res_img wrote (golden) and vs the oracle at full
    resolution; the drop-in export_res_img writes readable 16-bit files with the same content."""
    import os
    import PIL.Image as image
    from conftest import GOLDEN
    from neuralrgbd_amd import export_res
    g = dict(np.load(os.path.join(GOLDEN, "export_small.npz")))
    bv, img, d_candi = gen_golden.export_inputs()
    depth, conf, du, cu = export_res.depth_conf_u16(bv.to(DEV), d_candi)
    assert np.abs(depth.cpu().numpy() - g["depth"]).max() < 1e-5 and np.abs(conf.cpu().numpy() - g["conf"]).max() < 1e-6
    du_n, cu_n = du.cpu().numpy(), cu.cpu().numpy()
    # vs the reference's files: <= 1 LSB on a handful of pixels (torch.exp = sleef's 1-ulp expf and ATen's cascade sum there,
    # exp_rn and the sequential sum here; tests/test_oracle_golden.py has the count for the oracle) ...
    assert (np.abs(du_n.astype(np.int32) - g["depth_u16"].astype(np.int32)) > 1).sum() == 0
    nd, nc = int((du_n != g["depth_u16"]).sum()), int((cu_n != g["conf_u16"]).sum())
    print("[parity] export u16 vs the reference's .pgm files: depth %d / conf %d of %d pixels differ by 1 LSB" % (nd, nc, du_n.size))
    assert nd <= 4 and nc <= 4
    # ... and BIT-IDENTICAL to the oracle (bytes are bytes: same exp_rn operation sequence, same sequential sum)
    _, _, du_o, cu_o = co.export_depth_u16(bv[0].numpy(), d_candi)
    assert np.array_equal(du_n, du_o) and np.array_equal(cu_n, cu_o)
    export_res.export_res_img({"img": img}, bv.to(DEV), d_candi, str(tmp_path), 7)
    assert np.array_equal(np.array(image.open(str(tmp_path / "d_00007.pgm"))).astype(np.uint16), du_n)
    assert np.array_equal(np.array(image.open(str(tmp_path / "conf_00007.pgm"))).astype(np.uint16), cu_n)
    # full resolution, D = 64: bit-identical to the C oracle (same sequential sum)
    rng = np.random.RandomState(2)
    big = torch.log_softmax(torch.from_numpy(rng.standard_normal((64, 768, 1024)).astype(np.float32)) * 3, 0)
    d64 = np.linspace(0.1, 5.0, 64)
    from neuralrgbd_amd import ops
    dg, cg, dug, cug = ops.export_depth_u16(big.to(DEV), _dev(d64))
    do, co_, duo, cuo = co.export_depth_u16(big.numpy(), d64)
    assert np.array_equal(dg.cpu().numpy(), do) and np.array_equal(cg.cpu().numpy(), co_)
    assert np.array_equal(dug.cpu().numpy(), duo) and np.array_equal(cug.cpu().numpy(), cuo)
    # the whole fp32 range of exp_rn incl. subnormal results, through the D = 1 form of the kernel (depth = exp(x) * 1)
    x = np.concatenate([np.linspace(-104.5, 0, 300001), np.linspace(0, 89, 50001), [-np.inf, np.inf, np.nan]]).astype(np.float32)
    dg, _, _, _ = ops.export_depth_u16(_dev(x[None]), _dev(np.ones(1)))
    do, _, _, _ = co.export_depth_u16(x[None], np.ones(1))
    assert np.array_equal(dg.cpu().numpy(), do, equal_nan=True)


def test_pose_inverse_bit_exact_vs_oracle():
    """nrgbd_pose_inverse (fp64 Gauss-Jordan in a written-out order, rounded to fp32) == oracle_pose_inverse bit for bit:
    the PREDICT coordinates of the GPU path and of the oracle come from the SAME matrix (test_utils/test_KVNet.py:50)."""
    ops = _ops()
    rng = np.random.RandomState(5)
    T = np.stack([synth.random_pose(rng, 0.6, 2.0) for _ in range(1000)]).astype(np.float32)
    T[500:] += (rng.standard_normal((500, 4, 4)) * 1e-2).astype(np.float32)        # general (non-rigid) 4x4: pivoting paths
    T[10] = np.eye(4, dtype=np.float32)[[2, 0, 3, 1]]                               # permutation: every pivot is a swap
    got = ops.pose_inverse(_dev(T)).cpu().numpy()
    want = co.pose_inverse(T)
    assert np.array_equal(got, want)
    ex = np.linalg.inv(T.astype(np.float64))
    assert np.abs(got - ex).max() <= 0.51 * np.spacing(np.abs(ex).astype(np.float32)).max()
    # batch shapes, and one strided view as the host code passes it (poses[0, t_win_r])
    P = _dev(T[:8].reshape(2, 4, 4, 4))
    assert np.array_equal(ops.pose_inverse(P).cpu().numpy().reshape(8, 4, 4), want[:8])
    assert np.array_equal(ops.pose_inverse(P[1, 2]).cpu().numpy(), want[6])
    # singular input: NaN out + counted, no exception / sync on the device side
    S = T[:3].copy(); S[1, :, 2] = 0
    cnt = torch.zeros(1, dtype=torch.int32, device=DEV)
    out = ops.pose_inverse(_dev(S), singular_count=cnt).cpu().numpy()
    assert int(cnt.item()) == 1 and np.isnan(out[1]).all() and np.array_equal(out[[0, 2]], want[[0, 2]])


def test_resample_vol_cuda_with_new_candidates():
    """homography.resample_vol_cuda(..., d_candi_new=...) (the LBA driver's call, test_KVNet_LBA.py:414-417) against the
    oracle form that tests/test_oracle_vs_reference.py holds bit-identical to the live reference."""
    from neuralrgbd_amd import homography as H
    h, w, D = 24, 40, 16
    cam = camera.scannet_intrinsics(w, h)
    rng = np.random.RandomState(9)
    d_candi = np.linspace(0.3, 8, D)
    dpv = torch.log_softmax(torch.from_numpy(rng.standard_normal((1, D, h, w)).astype(np.float32)) * 4, 1)
    T = co.pose_inverse(synth.random_pose(rng, 0.05, 0.2).astype(np.float32))
    pad = math.log(1. / D)
    tan_hh, tan_hv = math.tan(math.radians(cam["hfov"]) * .5), math.tan(math.radians(cam["vfov"]) * .5)
    for d_new in (d_candi, np.linspace(0.5, 6.5, 11)):
        got = H.resample_vol_cuda(dpv.to(DEV), _dev(T), cam_intrinsic=cam, d_candi=d_candi, d_candi_new=d_new,
                                  padding_value=pad).cpu().numpy()
        d_pad = np.concatenate([d_new, np.zeros(D - len(d_new))])
        want = co.dpv_resample(dpv[0].numpy(), T, cam["unit_ray_array_2D"].numpy(), d_candi, tan_hh, tan_hv, pad,
                               clamp=None, d_candi_new=d_pad)
        assert got.shape == (D, h, w) and np.array_equal(got, want)
    with pytest.raises(IndexError):
        H.resample_vol_cuda(dpv.to(DEV), _dev(T), cam_intrinsic=cam, d_candi=d_candi, d_candi_new=np.linspace(1, 2, D + 1))
