"""K-Net kernels (conv3d.hip): fp32-MFMA 3x3x3 convolution with fused BatchNorm / ReLU / residual,
against torch's F.conv3d + F.batch_norm on the same GPU (a plain fp32 torch reference is the oracle for a
floating-point kernel) and against the CPU oracle for the whole stack."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _cl(x):   # [C,D,H,W] -> channels-last [D,H,W,C]
    return x.permute(1, 2, 3, 0).contiguous()


@pytest.mark.parametrize("D,H,W,Cin", [(4, 16, 32, 64), (2, 8, 16, 16), (5, 13, 21, 64), (3, 9, 40, 16), (8, 24, 48, 64)])
def test_conv3d_plain_vs_torch(D, H, W, Cin):
    from neuralrgbd_amd import ops
    g = torch.Generator().manual_seed(D * 1000 + H)
    x = torch.randn(Cin, D, H, W, generator=g).to(DEV)
    w = (torch.randn(64, Cin, 3, 3, 3, generator=g) * 0.05).to(DEV)
    want = F.conv3d(x[None], w, padding=1)[0]
    y, stats, _ = ops.conv3d(_cl(x), ops.conv3d_pack_weights(w))
    got = y.permute(3, 0, 1, 2)
    err = (got - want).abs().max().item()
    scale = want.abs().max().item()
    print("[parity] conv3d %dx%dx%d Cin=%d max|d|=%.3e (|y|max %.2f)" % (D, H, W, Cin, err, scale))
    assert err < 2e-5 * max(1.0, scale)
    # epilogue statistics = per-channel sum and sum of squares of the output
    s = stats.double().sum(0)
    assert torch.allclose(s[:64], want.double().sum((1, 2, 3)), rtol=1e-5, atol=1e-3)
    assert torch.allclose(s[64:], (want.double() ** 2).sum((1, 2, 3)), rtol=1e-5, atol=1e-3)


def test_conv3d_fused_prologue_and_materialize():
    """in = relu(x*s+t) + relu(res*s'+t'); zero padding applies to the ACTIVATED tensor."""
    from neuralrgbd_amd import ops
    D, H, W, C = 4, 10, 20, 64
    g = torch.Generator().manual_seed(5)
    x = torch.randn(C, D, H, W, generator=g).to(DEV)
    r = torch.randn(C, D, H, W, generator=g).to(DEV)
    w = (torch.randn(64, C, 3, 3, 3, generator=g) * 0.05).to(DEV)
    ss = torch.randn(C, 2, generator=g).to(DEV)
    rs = torch.randn(C, 2, generator=g).to(DEV)
    act = torch.relu(x * ss[:, 0, None, None, None] + ss[:, 1, None, None, None]) \
        + torch.relu(r * rs[:, 0, None, None, None] + rs[:, 1, None, None, None])
    want = F.conv3d(act[None], w, padding=1)[0]
    y, _, mat = ops.conv3d(_cl(x), ops.conv3d_pack_weights(w), x_ss=ss, x_relu=True, res=_cl(r), res_ss=rs,
                           res_relu=True, materialize=True)
    assert (y.permute(3, 0, 1, 2) - want).abs().max().item() < 2e-4
    assert (mat.permute(3, 0, 1, 2) - act).abs().max().item() < 1e-5
    # no-relu / identity residual form used by the residual blocks
    act2 = (x * ss[:, 0, None, None, None] + ss[:, 1, None, None, None]) + r
    y2, _, _ = ops.conv3d(_cl(x), ops.conv3d_pack_weights(w), x_ss=ss, res=_cl(r))
    assert (y2.permute(3, 0, 1, 2) - F.conv3d(act2[None], w, padding=1)[0]).abs().max().item() < 2e-4


def test_cout1_and_bn_finalize():
    from neuralrgbd_amd import ops
    D, H, W, C = 3, 11, 19, 64
    g = torch.Generator().manual_seed(6)
    x = torch.randn(C, D, H, W, generator=g).to(DEV)
    w1 = (torch.randn(1, C, 3, 3, 3, generator=g) * 0.05).to(DEV)
    ss = torch.randn(C, 2, generator=g).to(DEV)
    act = torch.relu(x * ss[:, 0, None, None, None] + ss[:, 1, None, None, None])
    want = F.conv3d(act[None], w1, padding=1)[0, 0]
    got = ops.conv3d_cout1(_cl(x), w1[0].reshape(C, 27).t().contiguous(), x_ss=ss, x_relu=True)
    assert (got - want).abs().max().item() < 1e-4
    # the depth-marching form: several depth chunks (the last one ragged), ragged plane tiles, one slice, no activation
    for (d2, h2, w2, act_on) in ((37, 24, 40, True), (20, 9, 16, False), (1, 8, 16, True), (9, 50, 70, True)):
        x2 = torch.randn(C, d2, h2, w2, generator=g).to(DEV)
        a2 = torch.relu(x2 * ss[:, 0, None, None, None] + ss[:, 1, None, None, None]) if act_on else x2
        want2 = F.conv3d(a2.double()[None], w1.double(), padding=1)[0, 0]
        got2 = ops.conv3d_cout1(_cl(x2), w1[0].reshape(C, 27).t().contiguous(), x_ss=ss if act_on else None, x_relu=act_on)
        assert got2.shape == (d2, h2, w2)
        assert (got2.double() - want2).abs().max().item() < 2e-5 * max(1.0, want2.abs().max().item()), (d2, h2, w2)
    # BatchNorm finalize: scale/shift reproduce F.batch_norm(training=True); running stats follow torch
    w = (torch.randn(64, C, 3, 3, 3, generator=g) * 0.05).to(DEV)
    y, stats, _ = ops.conv3d(_cl(x), ops.conv3d_pack_weights(w))
    gamma, beta = torch.rand(64, device=DEV) + 0.5, torch.randn(64, device=DEV)
    rm, rv = torch.zeros(64, device=DEV), torch.ones(64, device=DEV)
    rm_t, rv_t = rm.clone(), rv.clone()
    sc = ops.bn3d_finalize(stats, D * H * W, gamma, beta, 1e-5, 0.1, rm, rv)
    z = y.permute(3, 0, 1, 2)[None]
    want_bn = F.batch_norm(z, rm_t, rv_t, gamma, beta, training=True, momentum=0.1, eps=1e-5)
    got_bn = z * sc[:, 0].view(1, 64, 1, 1, 1) + sc[:, 1].view(1, 64, 1, 1, 1)
    assert (got_bn - want_bn).abs().max().item() < 1e-4
    assert torch.allclose(rm, rm_t, atol=1e-6) and torch.allclose(rv, rv_t, rtol=1e-5, atol=1e-6)


def test_knet_stack_vs_torch_modules():
    """forward_channels_last == the nn.Module forward (same weights) incl. the BatchNorm running-stat side effect."""
    import copy
    from neuralrgbd_amd import nets, synth
    torch.manual_seed(0)
    net = nets.KalmanGainNet(16, feature_dim=64)
    net.load_state_dict(synth.seeded_state_dict(net, 2))
    net = net.to(DEV)
    ref = copy.deepcopy(net)
    D, H, W = 8, 24, 32
    vol = torch.randn(1, 16, D, H, W, device=DEV)
    with torch.no_grad():
        want = ref(vol)[0, 0]
        got = net.forward_channels_last(vol[0].permute(1, 2, 3, 0).contiguous())
    err = (got - want).abs()
    print("[parity] K-Net stack (12 layers) max|d|=%.3e mean|d|=%.3e (|gain|max %.2f)" %
          (err.max().item(), err.mean().item(), want.abs().max().item()))
    assert err.max().item() < 2e-3 and err.mean().item() < 1e-4
    for (n1, b1), (n2, b2) in zip(net.named_buffers(), ref.named_buffers()):
        assert torch.allclose(b1.float(), b2.float(), rtol=1e-4, atol=1e-5), n1


@pytest.mark.parametrize("N,C,H,W,relu,res", [(5, 32, 24, 40, True, False), (5, 64, 16, 20, False, True), (1, 3, 8, 12, True, True)])
def test_bn2d_train_act_vs_torch(N, C, H, W, relu, res):
    from neuralrgbd_amd import ops
    g = torch.Generator().manual_seed(C + H)
    x = (torch.randn(N, C, H, W, generator=g) * 3 + 1).to(DEV)
    r = torch.randn(N, C, H, W, generator=g).to(DEV) if res else None
    gamma, beta = (torch.rand(C, generator=g) + 0.5).to(DEV), torch.randn(C, generator=g).to(DEV)
    want = F.batch_norm(x, None, None, gamma, beta, training=True, eps=1e-5)
    if relu:
        want = torch.relu(want)
    if res:
        want = want + r
    got, mv = ops.bn2d_train_act(x.clone(), gamma, beta, 1e-5, relu=relu, residual=r, want_mean_var=True)
    assert (got - want).abs().max().item() < 2e-5
    assert torch.allclose(mv[:, 0], x.mean((0, 2, 3)), atol=1e-5)
    assert torch.allclose(mv[:, 1], x.var((0, 2, 3), unbiased=False), rtol=1e-4, atol=1e-5)


def test_avgpool8_and_feature_cnn_fused_vs_modules():
    """The fused inference path of the feature CNN equals running the same modules the plain torch way."""
    import copy
    from neuralrgbd_amd import nets, ops, synth
    x = torch.randn(2, 7, 64, 128, device=DEV)
    assert (ops.avgpool8(x) - F.avg_pool2d(x, 8)).abs().max().item() < 1e-6
    net = nets.FeatureExtractor(feature_dim=64, multi_scale=True)
    net.load_state_dict(synth.seeded_state_dict(net, 4))
    net = net.to(DEV)
    ref = copy.deepcopy(net)
    img = torch.randn(5, 3, 256, 320, device=DEV)
    with torch.no_grad():
        half, feat = net(img)
    with torch.enable_grad():               # autograd on -> the plain module path
        half_r, feat_r = ref(img)
    e1, e2 = (half - half_r).abs().max().item(), (feat - feat_r).abs().max().item()
    print("[parity] feature CNN fused vs modules: layer1 max|d|=%.2e feat max|d|=%.2e (|feat|max %.1f)" %
          (e1, e2, feat_r.abs().max().item()))
    assert e1 < 1e-3 and e2 < 2e-3
    for (n1, b1), (n2, b2) in zip(net.named_buffers(), ref.named_buffers()):
        assert torch.allclose(b1.float(), b2.float(), rtol=1e-3, atol=1e-4), n1


# ----------------------------------------------------------------------------- generation 2: producer / consumer waves (wino_pc.hip)
@pytest.mark.parametrize("D,H,W", [(4, 16, 32), (2, 8, 16), (5, 13, 21), (3, 10, 40), (8, 24, 48), (1, 8, 16), (40, 40, 72)])
def test_conv_wino_pc_3d_plain_vs_torch(D, H, W):
    """Persistent producer/consumer Winograd kernel, kd = 3 (the K-Net's 64 -> 64 layers) vs F.conv3d in float64; the last
    grid has more tiles (900) than CUs, so workgroups walk several tiles and the weight ring wraps across them."""
    from neuralrgbd_amd import ops
    g = torch.Generator().manual_seed(D * 1000 + H)
    x = torch.randn(64, D, H, W, generator=g).to(DEV)
    w = (torch.randn(64, 64, 3, 3, 3, generator=g) * 0.05).to(DEV)
    want = F.conv3d(x[None].double(), w.double(), padding=1)[0]
    y, stats, _ = ops.conv_wino(_cl(x), ops.conv_wino_pack(w), 64, 3)
    err = (y.permute(3, 0, 1, 2).double() - want).abs().max().item()
    scale = want.abs().max().item()
    print("[parity] conv_wino_pc 3d %dx%dx%d max|d vs fp64|=%.3e (|y|max %.2f)" % (D, H, W, err, scale))
    assert err < 2e-5 * max(1.0, scale)
    assert stats.shape == (128, ops.conv_wino_tiles(D, H, W))
    s = stats.double().sum(1)
    assert torch.allclose(s[:64], want.sum((1, 2, 3)), rtol=1e-5, atol=1e-3)
    assert torch.allclose(s[64:], (want ** 2).sum((1, 2, 3)), rtol=1e-5, atol=1e-3)
    yd, _, _ = ops.conv3d(_cl(x), ops.conv3d_pack_weights(w))
    print("[parity] conv_wino_pc vs the direct kernel: max|d|=%.3e" % (y - yd).abs().max().item())
    assert (y - yd).abs().max().item() < 2e-5 * max(1.0, scale)


def test_conv_wino_pc_3d_fused_prologue_and_materialize():
    from neuralrgbd_amd import ops
    D, H, W, C = 6, 18, 36, 64
    g = torch.Generator().manual_seed(5)
    x = torch.randn(C, D, H, W, generator=g).to(DEV)
    r = torch.randn(C, D, H, W, generator=g).to(DEV)
    w = (torch.randn(64, C, 3, 3, 3, generator=g) * 0.05).to(DEV)
    ss = torch.randn(C, 2, generator=g).to(DEV)
    rs = torch.randn(C, 2, generator=g).to(DEV)
    act = torch.relu(x * ss[:, 0, None, None, None] + ss[:, 1, None, None, None]) \
        + torch.relu(r * rs[:, 0, None, None, None] + rs[:, 1, None, None, None])
    want = F.conv3d(act[None], w, padding=1)[0]
    wp = ops.conv_wino_pack(w)
    y, _, mat = ops.conv_wino(_cl(x), wp, 64, 3, x_ss=ss, x_relu=True, res=_cl(r), res_ss=rs, res_relu=True, materialize=True)
    assert (y.permute(3, 0, 1, 2) - want).abs().max().item() < 2e-4
    assert (mat.permute(3, 0, 1, 2) - act).abs().max().item() < 1e-5
    act2 = (x * ss[:, 0, None, None, None] + ss[:, 1, None, None, None]) + r
    y2, _, _ = ops.conv_wino(_cl(x), wp, 64, 3, x_ss=ss, res=_cl(r))
    assert (y2.permute(3, 0, 1, 2) - F.conv3d(act2[None], w, padding=1)[0]).abs().max().item() < 2e-4
    y3, _, _ = ops.conv_wino(_cl(x), wp, 64, 3, x_ss=ss, res=_cl(r))
    assert torch.equal(y2, y3)                                  # deterministic


def test_conv_wino_pack_device_vs_einsum():
    """nrgbd_conv_wino_pack (device, float64 inside) == the torch einsum reference, bit for bit, for every form in use."""
    from neuralrgbd_amd import ops
    g = torch.Generator().manual_seed(11)
    for shape in ((64, 64, 3, 3, 3), (128, 320, 3, 3), (64, 32, 3, 3), (128, 128, 3, 3)):
        w = torch.randn(*shape, generator=g).to(DEV)
        a, b = ops.conv_wino_pack(w), ops.conv_wino_pack_reference(w)
        assert a.shape == b.shape
        assert (a - b).abs().max().item() <= 1e-7 * b.abs().max().item(), shape
        if shape[1] % 64 == 0:             # the data-gradient stream straight from the forward weight == packing w^T flipped
            t = ops.conv_wino_pack(w, transposed=True)
            r = ops.conv_wino_pack_reference(w.transpose(0, 1).flip(*range(2, w.dim())).contiguous())
            assert (t - r).abs().max().item() <= 1e-7 * r.abs().max().item(), ("transposed", shape)


@pytest.mark.parametrize("D,H,W", [(4, 16, 32), (5, 13, 21), (40, 40, 72), (1, 8, 16)])
def test_conv_wino_pc_3d_first_layer_16_channels(D, H, W):
    """The K-Net's first layer (16 -> 64: 3 stages per tile, the odd-stage-count instantiation of the producer loop) vs
    F.conv3d in float64; the 900-tile grid makes workgroups walk several tiles, so the register-set parity flips per tile."""
    from neuralrgbd_amd import ops
    g = torch.Generator().manual_seed(D * 100 + W)
    x = torch.randn(16, D, H, W, generator=g).to(DEV)
    w = (torch.randn(64, 16, 3, 3, 3, generator=g) * 0.1).to(DEV)
    want = F.conv3d(x[None].double(), w.double(), padding=1)[0]
    y, stats, _ = ops.conv_wino(_cl(x), ops.conv_wino_pack(w), 64, 3)
    err = (y.permute(3, 0, 1, 2).double() - want).abs().max().item()
    print("[parity] conv_wino_pc 3d 16->64 %dx%dx%d max|d vs fp64|=%.3e (|y|max %.2f)" % (D, H, W, err, want.abs().max().item()))
    assert err < 2e-5 * max(1.0, want.abs().max().item())
    s = stats.double().sum(1)
    assert torch.allclose(s[:64], want.sum((1, 2, 3)), rtol=1e-5, atol=1e-3)
    assert torch.allclose(s[64:], (want ** 2).sum((1, 2, 3)), rtol=1e-5, atol=1e-3)


# ----------------------------------------------------------------------------- generation 3: Winograd along depth as well (wino_dw.hip)
@pytest.mark.parametrize("D,H,W,Cin", [(4, 16, 32, 64), (2, 8, 16, 64), (8, 24, 48, 64), (40, 40, 80, 64), (64, 8, 16, 64),
                                       (6, 16, 32, 16), (40, 40, 80, 16), (4, 16, 32, 128)])
def test_conv_wino_dw_plain_vs_torch(D, H, W, Cin):
    """F(2x2,3x3) in the plane + F(2,3) along depth (4 x Cin/16 stages per pair of slices) vs F.conv3d in float64, incl. grids with
    more tile pairs than CUs (persistent walk, ring and prefetch across tiles), a single pair (both depth borders in one tile),
    the one-stage-per-phase form (Cin = 16) and two channel groups of outputs; statistics rows; generation 2 as the A/B."""
    from neuralrgbd_amd import ops
    g = torch.Generator().manual_seed(D * 1000 + H + Cin)
    Cout = 128 if Cin == 128 else 64
    x = torch.randn(Cin, D, H, W, generator=g).to(DEV)
    w = (torch.randn(Cout, Cin, 3, 3, 3, generator=g) * 0.05).to(DEV)
    want = F.conv3d(x[None].double(), w.double(), padding=1)[0]
    assert ops.conv_wino_dw_supported(D, H, W, Cin, Cout)
    y, stats, _ = ops.conv_wino_dw(_cl(x), ops.conv_wino_dw_pack(w), Cout)
    err = (y.permute(3, 0, 1, 2).double() - want).abs()
    scale = want.abs().max().item()
    y2, _, _ = ops.conv_wino(_cl(x), ops.conv_wino_pack(w), Cout, 3)
    err2 = (y2.permute(3, 0, 1, 2).double() - want).abs()
    print("[parity] conv_wino_dw %dx%dx%d Cin=%d: max|d vs fp64| %.3e mean %.3e  (generation 2: %.3e / %.3e; |y|max %.2f)" %
          (D, H, W, Cin, err.max().item(), err.mean().item(), err2.max().item(), err2.mean().item(), scale))
    assert err.max().item() < 2e-5 * max(1.0, scale) and err.mean().item() < 2.0 * err2.mean().item() + 1e-9
    assert stats.shape == (2 * Cout, ops.conv_wino_tiles(D, H, W))
    s = stats.double().sum(1)
    assert torch.allclose(s[:Cout], want.sum((1, 2, 3)), rtol=1e-5, atol=1e-3)
    assert torch.allclose(s[Cout:], (want ** 2).sum((1, 2, 3)), rtol=1e-5, atol=1e-3)
    y3, _, _ = ops.conv_wino_dw(_cl(x), ops.conv_wino_dw_pack(w), Cout)
    assert torch.equal(y, y3)                                   # deterministic


def test_conv_wino_dw_fused_prologue_and_materialize():
    from neuralrgbd_amd import ops
    D, H, W, C = 6, 16, 32, 64
    g = torch.Generator().manual_seed(5)
    x = torch.randn(C, D, H, W, generator=g).to(DEV)
    r = torch.randn(C, D, H, W, generator=g).to(DEV)
    w = (torch.randn(64, C, 3, 3, 3, generator=g) * 0.05).to(DEV)
    ss = torch.randn(C, 2, generator=g).to(DEV)
    rs = torch.randn(C, 2, generator=g).to(DEV)
    act = torch.relu(x * ss[:, 0, None, None, None] + ss[:, 1, None, None, None]) \
        + torch.relu(r * rs[:, 0, None, None, None] + rs[:, 1, None, None, None])
    want = F.conv3d(act[None], w, padding=1)[0]
    wp = ops.conv_wino_dw_pack(w)
    y, _, mat = ops.conv_wino_dw(_cl(x), wp, 64, x_ss=ss, x_relu=True, res=_cl(r), res_ss=rs, res_relu=True, materialize=True)
    assert (y.permute(3, 0, 1, 2) - want).abs().max().item() < 2e-4
    assert (mat.permute(3, 0, 1, 2) - act).abs().max().item() < 1e-5
    act1 = torch.relu(x * ss[:, 0, None, None, None] + ss[:, 1, None, None, None])
    y1, _, _ = ops.conv_wino_dw(_cl(x), wp, 64, x_ss=ss, x_relu=True)
    assert (y1.permute(3, 0, 1, 2) - F.conv3d(act1[None], w, padding=1)[0]).abs().max().item() < 2e-4
    act2 = (x * ss[:, 0, None, None, None] + ss[:, 1, None, None, None]) + r
    y2, _, _ = ops.conv_wino_dw(_cl(x), wp, 64, x_ss=ss, res=_cl(r))
    assert (y2.permute(3, 0, 1, 2) - F.conv3d(act2[None], w, padding=1)[0]).abs().max().item() < 2e-4
    y3, _, _ = ops.conv_wino_dw(_cl(x), wp, 64, x_ss=ss, res=_cl(r))
    assert torch.equal(y2, y3)


def test_conv_wino_dw_pack_and_shape_contract():
    """Device packer == torch einsum reference (incl. the data-gradient form); unsupported shapes are refused, not substituted."""
    from neuralrgbd_amd import _lib, ops
    g = torch.Generator().manual_seed(13)
    for shape in ((64, 64, 3, 3, 3), (64, 16, 3, 3, 3), (128, 128, 3, 3, 3)):
        w = torch.randn(*shape, generator=g).to(DEV)
        a, b = ops.conv_wino_dw_pack(w), ops.conv_wino_dw_pack_reference(w)
        assert a.shape == b.shape and (a - b).abs().max().item() <= 1e-7 * b.abs().max().item(), shape
        if shape[1] % 64 == 0:
            t = ops.conv_wino_dw_pack(w, transposed=True)
            r = ops.conv_wino_dw_pack_reference(w.transpose(0, 1).flip(2, 3, 4).contiguous())
            assert (t - r).abs().max().item() <= 1e-7 * r.abs().max().item(), ("transposed", shape)
    w = ops.conv_wino_dw_pack(torch.randn(64, 64, 3, 3, 3, generator=g).to(DEV))
    for (D, H, W) in ((5, 16, 32), (4, 13, 32), (4, 16, 24)):
        assert not ops.conv_wino_dw_supported(D, H, W, 64, 64)
        with pytest.raises(_lib.NrgbdError):
            ops.conv_wino_dw(torch.zeros(D, H, W, 64, device=DEV), w, 64)


def test_knet_stack_dw_vs_the_other_kernels():
    """The whole K-Net on wino_dw.hip (what the path runs) vs the same stack on wino_pc.hip and on the direct kernel
    (`generation=`: a test-only argument): same graph, rounding order only."""
    from neuralrgbd_amd import nets
    torch.manual_seed(0)
    net = nets.KalmanGainNet(16, feature_dim=64).to(DEV)
    vol = torch.randn(8, 16, 32, 16, device=DEV)
    with torch.no_grad():
        a = net.forward_channels_last(vol, generation="wino_pc")
        b = net.forward_channels_last(vol)
        c = net.forward_channels_last(vol, generation="direct")
    print("[parity] K-Net stack: dw vs wino_pc max|d| %.3e (|gain| max %.2f); direct kernel vs wino_pc: %.3e" %
          ((a - b).abs().max().item(), a.abs().max().item(), (a - c).abs().max().item()))
    assert (a - b).abs().max().item() < 2e-4 * max(1.0, a.abs().max().item())
    assert (a - c).abs().max().item() < 2e-4 * max(1.0, a.abs().max().item())


@pytest.mark.parametrize("shape", [(64, 64, 3, 3), (128, 64, 3, 3), (128, 320 - 64, 3, 3), (64, 64, 3, 3, 3)])
def test_both_weight_streams_in_one_launch(shape):
    """transposed = 2 of the packing kernels (training): forward and data-gradient streams of a layer written by ONE launch are the
    streams the two single launches write (wino_pc.hip, and wino_dw.hip for 3-D weights)."""
    from neuralrgbd_amd import ops
    w = torch.randn(*shape, generator=torch.Generator().manual_seed(sum(shape))).to(DEV)
    f, b = ops.conv_wino_pack_both(w)
    assert torch.equal(f, ops.conv_wino_pack(w)) and torch.equal(b, ops.conv_wino_pack(w, True))
    if w.dim() == 5:
        f, b = ops.conv_wino_pack_both(w, dw=True)
        assert torch.equal(f, ops.conv_wino_dw_pack(w)) and torch.equal(b, ops.conv_wino_dw_pack(w, True))
        f, b = ops.conv_wino_pack_both(w, dw=4)
        assert torch.equal(f, ops.conv_wino_dw4_pack(w)) and torch.equal(b, ops.conv_wino_dw4_pack(w, True))
        # the data-gradient stream = the forward stream of the transposed + flipped weights
        assert torch.equal(b, ops.conv_wino_dw4_pack(w.transpose(0, 1).flip(2, 3, 4).contiguous()))
    with pytest.raises(ValueError):
        ops.conv_wino_pack(torch.zeros(64, 32, 3, 3, device=DEV), transposed=2)      # 32 inputs cannot be a 64-column output group


def test_wino_dw_clamped_fma_relu_has_the_bits_of_the_plain_form():
    """nrgbd_conv_wino_dw_unit_f32: relu(x * s + t) taken by the producers' FMA clamp on operands scaled by 2^-k, weights packed
    from 2^k * w — output and statistics bit-identical to the plain form (v_max_f32 per element), for a bound just above the
    largest activation and for a generous one; ops.relu_unit bounds a BatchNorm's output whatever the data."""
    from neuralrgbd_amd import ops
    D, H, W, C = 8, 24, 48, 64
    g = torch.Generator().manual_seed(21)
    x = (torch.randn(D, H, W, C, generator=g) * 3.0).to(DEV)
    w = (torch.randn(C, C, 3, 3, 3, generator=g) * 0.05).to(DEV)
    ss = torch.randn(C, 2, generator=g).to(DEV)
    y0, st0, _ = ops.conv_wino_dw(x, ops.conv_wino_dw_pack(w), C, x_ss=ss, x_relu=True)
    act_max = torch.relu(x * ss[:, 0] + ss[:, 1]).max().item()
    import math
    for k in (math.floor(math.log2(act_max)) + 1, 14):
        y1, st1, _ = ops.conv_wino_dw(x, ops.conv_wino_dw_pack(w * 2.0 ** k), C, x_ss=ss, x_relu=True, x_unit=2.0 ** -k)
        assert torch.equal(y0, y1) and torch.equal(st0, st1), k
    with pytest.raises(Exception):
        ops.conv_wino_dw(x, ops.conv_wino_dw_pack(w), C, x_ss=ss, x_relu=True, x_unit=0.3)      # not a power of two
    # the bound: a BatchNorm output with batch statistics never exceeds |gamma| sqrt(n) + |beta|, even for one huge outlier
    n = 4096
    yv = torch.zeros(n, 4); yv[0] = 1e6
    gamma, beta = torch.tensor([1.5, -0.3, 2.0, 0.1]), torch.tensor([0.2, -4.0, 0.0, 3.0])
    z = torch.nn.functional.batch_norm(yv, None, None, gamma, beta, training=True, eps=1e-5)
    assert z.abs().max().item() < 1.0 / ops.relu_unit(gamma, beta, n)


def test_variance_collapse_is_loud_not_saturated():
    """the finalisers take the variance as E[y^2] - mean^2 from fp32 per-tile partials; a channel with
    std / |mean| below ~3e-3 has no correct digit left, and the clamped-FMA ReLU of the next K-Net layer (ops.relu_unit assumes a
    computed variance >= true / 4) could saturate silently.  Such a channel must get a NaN scale (csrc/common.hpp
    bn_finalize_channel) so that everything it feeds is NaN: (1) crafted partials through all three finalisers, (2) real
    statistics of a collapsing map through nhwc_stats -> bn_finalize (a healthy channel with |mean| / std = 100 stays finite and
    accurate), (3) the kernels' ReLU forms turn a NaN scale into a switched-off channel — the NaN alone would vanish —, which is
    why (5) the finalisers also count the collapse into a status word that raises NrgbdError at the path's synchronisation points."""
    from neuralrgbd_amd import ops
    C, rows, count = 64, 48, 48 * 256
    g = torch.Generator().manual_seed(5)
    mean = torch.full((C,), 50.0)
    std = torch.ones(C)
    std[3], std[17] = 0.05, 1e-4                      # std / |mean| = 1e-3 and 2e-6: collapsed
    mean[40], std[40] = 0.0, 0.0                      # a dead channel (all zero): NOT an error, scale finite
    s1 = (mean * count / rows)[:, None].expand(C, rows)
    s2 = ((std ** 2 + mean ** 2) * count / rows)[:, None].expand(C, rows)
    cm = torch.cat((s1, s2), 0).contiguous().to(DEV)                      # column-major partials [2C, rows]
    rm = torch.cat((s1.t(), s2.t()), 1).contiguous().to(DEV)              # row-major partials [rows, 2C]
    gamma, beta = torch.ones(C, device=DEV), torch.zeros(C, device=DEV)
    word = torch.zeros(1, dtype=torch.int32, device=DEV)
    for i, fin in enumerate((lambda: ops.bn_finalize_cm(cm, count, gamma, beta, 1e-5, 0.1, status=word),
                             lambda: ops.bn_finalize(rm, count, gamma, beta, 1e-5, 0.1, status=word),
                             lambda: ops.bn3d_finalize(rm, count, gamma, beta, 1e-5, 0.1, status=word))):
        ss = fin()
        bad = torch.isnan(ss[:, 0]).cpu()
        assert bad[3] and bad[17] and int(bad.sum()) == 2, bad.nonzero().flatten().tolist()
        assert torch.isfinite(ss[40]).all()
        assert int(word.item()) == 2 * (i + 1)                            # counted into the caller's status word
    # (2) measured statistics: x = 1000 + 1e-3 N(0,1) in channel 5, N(0,1) + 100 in channel 6 (healthy, |mean|/std = 100)
    x = torch.randn(8, 24, 48, C, generator=g)
    x[..., 5] = 1000.0 + 1e-3 * x[..., 5]
    x[..., 6] = 100.0 + x[..., 6]
    x = x.to(DEV)
    ss = ops.bn_finalize(ops.nhwc_stats(x), x.numel() // C, gamma, beta, 1e-5, 0.1)
    assert torch.isnan(ss[5, 0]) and int(torch.isnan(ss[:, 0]).sum()) == 1
    want6 = 1.0 / x[..., 6].double().std(unbiased=False).item()
    assert abs(ss[6, 0].item() - want6) < 2e-3 * want6                    # E[y^2] - mean^2 at |mean| / std = 100: 3 digits, finite
    # (3) WHY the status word exists: the kernels' ReLU maps NaN to 0 (v_max_f32 returns its non-NaN operand; the FMA clamp of the
    # clamped form maps NaN to 0 under the kernels' DX10_CLAMP mode), so the NaN scale of a collapsed channel silently becomes
    # "channel switched off" in the next convolution — finite, plausible, wrong.  Both forms of the K-Net layer show it:
    w = (torch.randn(C, C, 3, 3, 3, generator=g) * 0.05).to(DEV)
    unit = ops.relu_unit(gamma, beta, x.numel() // C)
    ss_off = ss.clone(); ss_off[5] = 0.0
    for kw, wp in ((dict(x_unit=unit), ops.conv_wino_dw_pack(w / unit)), (dict(), ops.conv_wino_dw_pack(w))):
        y, _, _ = ops.conv_wino_dw(x, wp, C, x_ss=ss, x_relu=True, **kw)
        y_off, _, _ = ops.conv_wino_dw(x, wp, C, x_ss=ss_off, x_relu=True, **kw)
        assert torch.isfinite(y).all() and torch.equal(y, y_off)
    # (4) a non-ReLU consumer (every residual block's second BatchNorm) does propagate it
    assert torch.isnan(ops.nhwc_act(x, ss, False)[..., 5]).all()
    # (5) the host mirror is LOUD: the status word raises at the path's own synchronisation points
    import neuralrgbd_amd
    from neuralrgbd_amd import camera, misc, nets, synth
    from neuralrgbd_amd._lib import NrgbdError
    from neuralrgbd_amd.streaming import DepthStream
    dev = torch.device(DEV)
    nets.check_status(dev)                                                # clean
    nets.status_word(dev).fill_(3)
    with pytest.raises(NrgbdError, match="collapsed in 3 channel"):
        misc.valid_dpv(torch.zeros(1, 4, 2, 2, device=DEV))              # KVNET.forward's probe of BV_predict
    assert int(nets.status_word(dev).item()) == 0                         # cleared by the raise
    H, W, D = 256, 256, 16
    cam = camera.scannet_intrinsics(W // 4, H // 4)
    d_candi = np.linspace(0.1, 5.0, D)
    model = neuralrgbd_amd.KVNET(64, cam, d_candi, 10.0, 64, None, if_refined=True, refineNet_name="DPV", t_win_r=2)
    model.load_state_dict(synth.seeded_state_dict(model, 0))
    st = DepthStream(model.to(DEV), cam, d_candi, use_graph=False)
    r, s_, p_ = (t.to(DEV) for t in synth.noise_window(5, H, W))
    st.step(r, s_, p_); st.step(r, s_, p_)
    st.check()                                                            # healthy frames report nothing
    nets.status_word(dev).fill_(1)                                        # what a finaliser of the next frame would have done
    st.step(r, s_, p_)                                                    # queues the 4-byte copy behind the frame
    torch.cuda.synchronize()
    with pytest.raises(NrgbdError):
        st.step(r, s_, p_)                                                # ... and the following step sees it


@pytest.mark.parametrize("D,H,W,Cin", [(4, 16, 32, 64), (8, 24, 48, 64), (40, 40, 80, 64), (64, 8, 16, 64), (8, 16, 32, 16), (4, 16, 32, 128)])
def test_conv_wino_dw4_plain_vs_torch(D, H, W, Cin):
    """csrc/wino_dw4.hip: F(2x2,3x3) in the plane + F(4,3) along depth (6 x Cin/16 stages per FOUR output slices; points 0, +-1/2,
    +-3/2, inf) vs F.conv3d in float64: a single quadruple (both depth borders in one tile: the empty first and last input slices),
    grids with more tiles than CUs, one stage per phase (Cin = 16), two output channel groups; statistics rows; wino_dw as the A/B."""
    from neuralrgbd_amd import ops
    g = torch.Generator().manual_seed(D * 1000 + H + Cin)
    Cout = 128 if Cin == 128 else 64
    x = torch.randn(Cin, D, H, W, generator=g).to(DEV)
    w = (torch.randn(Cout, Cin, 3, 3, 3, generator=g) * 0.05).to(DEV)
    want = F.conv3d(x[None].double(), w.double(), padding=1)[0]
    assert ops.conv_wino_dw4_supported(D, H, W, Cin, Cout)
    y, stats = ops.conv_wino_dw4(_cl(x), ops.conv_wino_dw4_pack(w), Cout)
    err = (y.permute(3, 0, 1, 2).double() - want).abs()
    scale = want.abs().max().item()
    y2, _, _ = ops.conv_wino_dw(_cl(x), ops.conv_wino_dw_pack(w), Cout)
    err2 = (y2.permute(3, 0, 1, 2).double() - want).abs()
    print("[parity] conv_wino_dw4 %dx%dx%d Cin=%d: max|d vs fp64| %.3e mean %.3e  (wino_dw: %.3e / %.3e; |y|max %.2f)" %
          (D, H, W, Cin, err.max().item(), err.mean().item(), err2.max().item(), err2.mean().item(), scale))
    assert err.max().item() < 4e-5 * max(1.0, scale) and err.mean().item() < 3.0 * err2.mean().item() + 1e-9
    assert stats.shape == (2 * Cout, ops.conv_wino_tiles(D, H, W))
    s = stats.double().sum(1)
    assert torch.allclose(s[:Cout], want.sum((1, 2, 3)), rtol=1e-5, atol=2e-3)
    assert torch.allclose(s[Cout:], (want ** 2).sum((1, 2, 3)), rtol=1e-5, atol=2e-3)
    y3, _ = ops.conv_wino_dw4(_cl(x), ops.conv_wino_dw4_pack(w), Cout)
    assert torch.equal(y, y3)                                   # deterministic


def test_conv_wino_dw4_input_forms_and_clamped_relu():
    """The three input forms of wino_dw4.hip against wino_dw.hip's on the same operands (same activation arithmetic, only the depth
    transform differs): act(x * s + t) with and without ReLU, and the clamped-FMA ReLU — bit-identical to its own plain form."""
    from neuralrgbd_amd import ops
    D, H, W, C = 8, 24, 48, 64
    g = torch.Generator().manual_seed(33)
    x = (torch.randn(D, H, W, C, generator=g) * 3.0).to(DEV)
    w = (torch.randn(C, C, 3, 3, 3, generator=g) * 0.05).to(DEV)
    ss = torch.randn(C, 2, generator=g).to(DEV)
    wp4, wp2 = ops.conv_wino_dw4_pack(w), ops.conv_wino_dw_pack(w)
    for relu in (False, True):
        y4, st4 = ops.conv_wino_dw4(x, wp4, C, x_ss=ss, x_relu=relu)
        y2, st2, _ = ops.conv_wino_dw(x, wp2, C, x_ss=ss, x_relu=relu)
        sc = y2.abs().max().item()
        assert (y4 - y2).abs().max().item() < 3e-5 * max(1.0, sc), relu
        assert torch.allclose(st4.double().sum(1), st2.double().sum(1), rtol=1e-4, atol=1e-2)
    y0, st0 = ops.conv_wino_dw4(x, wp4, C, x_ss=ss, x_relu=True)
    import math
    act_max = torch.relu(x * ss[:, 0] + ss[:, 1]).max().item()
    for k in (math.floor(math.log2(act_max)) + 1, 14):
        y1, st1 = ops.conv_wino_dw4(x, ops.conv_wino_dw4_pack(w * 2.0 ** k), C, x_ss=ss, x_relu=True, x_unit=2.0 ** -k)
        assert torch.equal(y0, y1) and torch.equal(st0, st1), k
    with pytest.raises(Exception):
        ops.conv_wino_dw4(x[:6], wp4, C)                                     # D % 4 != 0
