"""Parity at BASELINE.json's full size (config B: plane-sweep grid 192 x 256, D = 64, V = 4, C = 67), where the
CPU oracle is too slow to run in a test: size-independent properties and cross-checks between independent
implementations on the GPU."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from neuralrgbd_amd import camera, ops, synth

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
H_, W_, D_, V_ = 192, 256, 64, 4


def _setup(seed=0):
    from neuralrgbd_amd import homography as Hm, ops
    cam = camera.scannet_intrinsics(W_, H_)
    rng = np.random.RandomState(seed)
    feats = torch.from_numpy(rng.standard_normal((V_ + 1, 64, H_, W_)).astype(np.float32)).to(DEV)
    frames = torch.from_numpy(rng.standard_normal((V_ + 1, 3, 4 * H_, 4 * W_)).astype(np.float32)).to(DEV)
    poses = torch.from_numpy(synth.random_poses(rng, V_)).to(DEV)
    K, rays = Hm._cam_dev(cam, torch.device(DEV))
    d = Hm._d_candi_dev(np.linspace(0.1, 5.0, D_), torch.device(DEV))
    KR, Kt = Hm.homography_terms(K, poses[:, :3, :3], poses[:, :3, 3])
    tex = ops.pack_nhwc(feats, frames)
    return cam, tex, KR, Kt, rays, d, poses


def test_three_kernel_generations_agree_and_are_deterministic():
    from neuralrgbd_amd import ops
    cam, tex, KR, Kt, rays, d, _ = _setup()
    args = (tex[V_], tex[:V_], KR, Kt, rays, d, W_ / 2.0, H_ / 2.0, 10.0, 67)
    c1, l1 = ops.costvol(*args, want_cost=True, want_logp=True, generation="quad")
    c1b, _ = ops.costvol(*args, want_cost=True, want_logp=False, generation="quad")
    c2, l2 = ops.costvol(*args, want_cost=True, want_logp=True, generation="gather")
    c3, _ = ops.costvol(*args, want_cost=True, want_logp=False, generation="lds")
    assert (c3 - c2).abs().max().item() < 1e-4 * max(1.0, c1.abs().max().item() / 10)
    assert torch.equal(c1, c1b)                                         # no races: bitwise reproducible
    err = (c1 - c2).abs().max().item()
    print("[parity] full-size costvol: quad generation vs gather generation max|d|=%.2e (cost up to %.1f)" % (err, c1.max().item()))
    assert err < 1e-4 * max(1.0, c1.abs().max().item() / 10)
    assert (l1 - l2).abs().max().item() < 2e-4
    # arg-max depth index: on pure-noise features ~0.25 % of the 49,152 pixels have their two best candidates within
    # fp32 summation noise of each other; every disagreement must be such a near tie (gap < 1e-3 in log-prob)
    from conftest import near_tie_mismatches
    raw = int((l1.argmax(0) != l2.argmax(0)).sum())
    real = near_tie_mismatches(l1.cpu().numpy(), l2.cpu().numpy(), 1e-3)
    print("[parity] full-size arg-max: %d/%d differ between generations, %d not explained by a near tie" % (raw, H_ * W_, real))
    assert real == 0 and raw < 0.01 * H_ * W_
    # log-softmax over D: every pixel's probabilities sum to one
    assert (torch.logsumexp(l1.double(), dim=0)).abs().max().item() < 1e-5


def test_identity_motion_gives_zero_cost_everywhere():
    from neuralrgbd_amd import ops
    cam, tex, KR, Kt, rays, d, poses = _setup(1)
    from neuralrgbd_amd import homography as Hm
    K, _ = Hm._cam_dev(cam, torch.device(DEV))
    eye = torch.eye(4, device=DEV).expand(V_, 4, 4).contiguous()
    KR0, Kt0 = Hm.homography_terms(K, eye[:, :3, :3], eye[:, :3, 3])
    same = tex[V_:V_ + 1].expand(V_, H_, W_, 68).contiguous()
    cost, _ = ops.costvol(tex[V_], same, KR0, Kt0, rays, d, W_ / 2.0, H_ / 2.0, 1.0, 67)
    assert cost.max().item() < 1e-5


def test_predict_identity_and_bounds():
    """PREDICT: the clamp bounds hold at full size and the output never exceeds the input's range + the pad value."""
    from neuralrgbd_amd import homography as Hm
    cam, tex, KR, Kt, rays, d, poses = _setup(2)
    dpv = torch.log_softmax(torch.randn(1, D_, H_, W_, device=DEV) * 4, dim=1)
    pad = float(np.log(1.0 / D_))
    out = Hm.resample_vol_cuda(dpv, ops.pose_inverse(poses[2].contiguous()), cam_intrinsic=cam, d_candi=np.linspace(0.1, 5, D_),
                               padding_value=pad, clamp=(-1000., 0.))
    assert out.shape == (D_, H_, W_) and bool(torch.isfinite(out).all())
    assert out.max().item() <= 0.0 and out.min().item() >= min(dpv.min().item(), pad) - 1e-4


def test_conv3d_full_size_vs_vendor_convolution():
    from neuralrgbd_amd import ops
    g = torch.Generator().manual_seed(3)
    x = torch.randn(64, 16, H_, W_, generator=g).to(DEV)               # 16 depth slices of the full grid
    w = (torch.randn(64, 64, 3, 3, 3, generator=g) * 0.04).to(DEV)
    want = F.conv3d(x[None], w, padding=1)[0]
    y, _, _ = ops.conv3d(x.permute(1, 2, 3, 0).contiguous(), ops.conv3d_pack_weights(w))
    err = (y.permute(3, 0, 1, 2) - want).abs().max().item()
    print("[parity] full-grid conv3d vs MIOpen: max|d|=%.2e (|y|max %.1f)" % (err, want.abs().max().item()))
    assert err < 3e-5 * max(1.0, want.abs().max().item())


def test_stream_graph_replay_equals_eager():
    """DepthStream: hipGraph-replayed frames reproduce eager launches (the vendor convolutions may pick a different
    algorithm under capture, so equality is to fp32 summation noise, not bitwise)."""
    import neuralrgbd_amd
    from neuralrgbd_amd.streaming import DepthStream
    H, W, D = 256, 384, 16
    cam = camera.scannet_intrinsics(W // 4, H // 4)
    d_candi = np.linspace(0.1, 5, D)
    outs = []
    for use_graph in (False, True):
        model = neuralrgbd_amd.KVNET(64, cam, d_candi, 10.0, 64, None, if_refined=True, refineNet_name="DPV", t_win_r=2)
        model.load_state_dict(synth.seeded_state_dict(model, 0))
        stream = DepthStream(model.to(DEV), cam, d_candi, use_graph=use_graph)
        last = None
        for i in range(5):
            r, s, p = (t.to(DEV) for t in synth.noise_window(70 + i, H, W))
            refined, dpv = stream.step(r, s, p)
            last = (refined.clone(), dpv.clone(), stream.bv_predict.clone())
        assert (stream._graph is not None) == use_graph, stream.graph_error
        outs.append(last)
    for name, a, b in zip(("R(DPV)", "DPV", "BV_predict"), *outs):
        err = (a - b).abs()
        print("[parity] graph replay vs eager %-10s max|d|=%.2e mean|d|=%.2e" % (name, err.max().item(), err.mean().item()))
        assert err.mean().item() < 1e-4 and err.max().item() < 5e-3


@pytest.mark.parametrize("H,W,D", [(256, 384, 64), (256, 256, 16)])
def test_pipelined_stream_is_bit_identical_to_the_sequential_one(H, W, D):
    """DepthStream(pipeline=True): the D-Net of frame t + 1 on a second HIP stream under the K-Net / R-Net / PREDICT of frame t.
    Same kernels on the same inputs: every frame's refined DPV, DPV and predicted state equal the sequential stream's BIT FOR BIT
    (eager warm-up frames, the capture frame, replays of both slots), one call later; flush() hands out the last frame."""
    import neuralrgbd_amd
    from neuralrgbd_amd.streaming import DepthStream
    cam = camera.scannet_intrinsics(W // 4, H // 4)
    d_candi = np.linspace(0.1, 5, D)
    wins = [tuple(t.to(DEV) for t in synth.noise_window(90 + i, H, W)) for i in range(10)]
    got = {}
    for pipe in (False, True):
        model = neuralrgbd_amd.KVNET(64, cam, d_candi, 10.0, 64, None, if_refined=True, refineNet_name="DPV", t_win_r=2)
        model.load_state_dict(synth.seeded_state_dict(model, 0))
        stream = DepthStream(model.to(DEV), cam, d_candi, use_graph=True, pipeline=pipe, copy_outputs=True)
        outs = []
        for i, (r, s, p) in enumerate(wins):
            o = stream.step(r, s, p)
            if o is not None:
                outs.append((o[0].clone(), o[1].clone()))
        if pipe:
            assert len(outs) == len(wins) - 1            # one frame of latency ...
            outs.append(stream.flush())                  # ... handed out here
            assert stream.flush() is None
        assert stream._graph is not None, stream.graph_error
        torch.cuda.synchronize()
        got[pipe] = (outs, stream.bv_predict.clone())
    assert len(got[True][0]) == len(got[False][0]) == len(wins)
    for i, (a, b) in enumerate(zip(got[True][0], got[False][0])):
        assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]), "frame %d" % i
    assert torch.equal(got[True][1], got[False][1])
