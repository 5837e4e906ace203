"""R-Net (models/Refine.py:24-107) on the hand-written matrix-core kernels: every layer form against the torch operator it
replaces (fp32), then the whole up-sampler against the nn.Module graph, both channel layouts of the image features."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from neuralrgbd_amd import synth

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _rand(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(DEV)


@pytest.mark.parametrize("H,W,Cin,Cout,cin_pad,cout_pad", [(32, 48, 128, 128, 128, 128), (40, 56, 96, 96, 96, 96),
                                                            (33, 50, 67, 67, 80, 96), (48, 64, 67, 64, 80, 64)])
def test_conv3x3_bias_lrelu_into_wider_pixels(H, W, Cin, Cout, cin_pad, cout_pad):
    """mode 0: conv2d_leakyRelu with zero-padded channels, written at a channel offset of a wider buffer."""
    from neuralrgbd_amd import ops
    x = _rand(1, Cin, H, W, seed=1)
    w = _rand(Cout, Cin, 3, 3, seed=2, scale=0.05)
    b = _rand(Cout, seed=3, scale=0.1)
    want = F.leaky_relu(F.conv2d(x, w, b, 1, 1), 0.01)
    xp = torch.zeros(1, H, W, cin_pad, device=DEV)
    xp[..., :Cin] = x.permute(0, 2, 3, 1)
    wp = torch.zeros(cout_pad, cin_pad, 3, 3, device=DEV)
    wp[:Cout, :Cin] = w
    bp = torch.zeros(cout_pad, device=DEV)
    bp[:Cout] = b
    ldy, off = Cout + 21, 8
    out = torch.full((1, H, W, ldy), 7.0, device=DEV)
    ops.conv2d_rnet(xp, ops.conv_pack_weights(wp), cout_pad, bias=bp, out=out, ldy=ldy, ycoff=off, cout_valid=Cout)
    got = out[..., off:off + Cout].permute(0, 3, 1, 2)
    err = (got - want).abs().max().item()
    print("[parity] R-Net conv %d->%d %dx%d: max|d|=%.2e (|y|max %.1f)" % (Cin, Cout, H, W, err, want.abs().max().item()))
    assert err < 2e-5 * max(1.0, want.abs().max().item())
    assert bool((out[..., :off] == 7.0).all()) and bool((out[..., off + Cout:] == 7.0).all())   # nothing else touched


@pytest.mark.parametrize("H,W,Cin", [(24, 40, 128), (33, 29, 96)])
def test_transposed_conv_as_four_subpixel_phases(H, W, Cin):
    """mode 1: ConvTranspose2d(k4, s2, p1) + bias + LeakyReLU = four 2x2-tap launches."""
    from neuralrgbd_amd import ops
    x = _rand(1, Cin, H, W, seed=4)
    w = _rand(Cin, 64, 4, 4, seed=5, scale=0.05)
    b = _rand(64, seed=6, scale=0.1)
    want = F.leaky_relu(F.conv_transpose2d(x, w, b, 2, 1), 0.01)
    xc = x.permute(0, 2, 3, 1).contiguous()
    out = torch.zeros(1, 2 * H, 2 * W, 80, device=DEV)
    packed = []
    for pa in (0, 1):
        for pb in (0, 1):
            ky = [3, 1] if pa == 0 else [2, 0]
            kx = [3, 1] if pb == 0 else [2, 0]
            wph = w[:, :, ky][:, :, :, kx].permute(1, 0, 2, 3).contiguous()
            packed.append(ops.conv_pack_weights(wph))
            ops.conv2d_rnet(xc, packed[-1], 64, bias=b, out=out, ldy=80, ycoff=0, cout_valid=64, mode=1, pa=pa, pb=pb)
    got = out[..., :64].permute(0, 3, 1, 2)
    err = (got - want).abs().max().item()
    print("[parity] R-Net transposed conv %d->64 %dx%d: max|d|=%.2e" % (Cin, H, W, err))
    assert err < 2e-5 * max(1.0, want.abs().max().item())
    assert bool((out[..., 64:] == 0).all())
    # mode 3: the same four phases as ONE launch -> bit-identical
    out3 = torch.zeros_like(out)
    ops.conv2d_rnet(xc, torch.cat(packed), 64, bias=b, out=out3, ldy=80, ycoff=0, cout_valid=64, mode=3)
    assert torch.equal(out3, out)


@pytest.mark.parametrize("H,W", [(32, 48), (37, 45)])
def test_last_conv_with_planar_log_softmax(H, W):
    """mode 2: conv2_2 + bias + F.log_softmax(dim=1), stored [1,D,H,W]."""
    from neuralrgbd_amd import ops
    x = _rand(1, 64, H, W, seed=7)
    w = _rand(64, 64, 3, 3, seed=8, scale=0.08)
    b = _rand(64, seed=9, scale=0.1)
    want = F.log_softmax(F.conv2d(x, w, b, 1, 1), dim=1)
    got = ops.conv2d_rnet(x.permute(0, 2, 3, 1).contiguous(), ops.conv_pack_weights(w), 64, bias=b, lrelu=False, mode=2)
    assert got.shape == want.shape and got.is_contiguous()
    err = (got - want).abs().max().item()
    print("[parity] R-Net conv2_2 + log-softmax %dx%d: max|d|=%.2e" % (H, W, err))
    assert err < 3e-5 and int((got.argmax(1) != want.argmax(1)).sum()) == 0


@pytest.mark.parametrize("channels_last_feats,D", [(False, 64), (True, 64), (True, 128), (True, 16), (False, 8), (True, 24), (True, 32), (True, 100)])
def test_whole_rnet_matrix_core_path_vs_modules(channels_last_feats, D):
    """D = 64 (configs S, B, K) and D = 128 (config H: 192 / 160 / 131-channel layers as output-column slices); any other
    candidate count zero-padded to the next of the two (round 5: D = 16 — the reference-golden stream and smoke() — used to
    take MIOpen's convolutions)."""
    from neuralrgbd_amd import nets
    h, w = 48, 64
    net = nets.DPVUpsampleNet(64, 32, 3, D=D)
    net.load_state_dict(synth.seeded_state_dict(net, 3))
    net = net.to(DEV)
    dpv_log = torch.log_softmax(_rand(1, D, h, w, seed=10, scale=3.0), dim=1)
    feats = [_rand(1, 64, h, w, seed=11), _rand(1, 32, 2 * h, 2 * w, seed=12), _rand(1, 3, 4 * h, 4 * w, seed=13)]
    if channels_last_feats:   # what the matrix-core feature trunk hands over: NCHW-shaped views of channels-last tensors
        feats = [f.permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2) for f in feats]
    import copy
    with torch.no_grad():
        ref = copy.deepcopy(net).cpu().double()                 # the plain nn.Module graph (Refine.py:79-107) in float64 on the host
        mod = ref(torch.exp(dpv_log).cpu().double(), [f.contiguous().cpu().double() for f in feats]).float().to(DEV)
        assert net.mfma_ok(dpv_log)
        got = net.forward_log(dpv_log, feats)
        got2 = net.forward_log(dpv_log, feats)
    assert torch.equal(got, got2)
    e1 = (got - mod).abs()
    print("[parity] whole R-Net: matrix-core path vs the module graph in float64: max %.2e mean %.2e" % (e1.max().item(), e1.mean().item()))
    assert e1.mean().item() < 1e-5 and e1.max().item() < 2e-4
    assert int((got.argmax(1) != mod.argmax(1)).sum()) <= 2


def test_batch_of_two_equals_two_calls():
    """KVNET.forward refines BV_cur and DPV as one batch of 2 in the update branch: identical to two single calls."""
    from neuralrgbd_amd import nets
    h, w, D = 32, 48, 64
    net = nets.DPVUpsampleNet(64, 32, 3, D=D)
    net.load_state_dict(synth.seeded_state_dict(net, 4))
    net = net.to(DEV)
    a = torch.log_softmax(_rand(1, D, h, w, seed=20, scale=3.0), dim=1)
    b = torch.log_softmax(_rand(1, D, h, w, seed=21, scale=2.0), dim=1)
    feats = [_rand(1, 64, h, w, seed=22), _rand(1, 32, 2 * h, 2 * w, seed=23), _rand(1, 3, 4 * h, 4 * w, seed=24)]
    with torch.no_grad():
        ra, rb = net.forward_log(a, feats).clone(), net.forward_log(b, feats).clone()
        both = net.forward_log(torch.cat((a, b), 0), feats)
    assert torch.equal(both[0:1], ra) and torch.equal(both[1:2], rb)
    with torch.no_grad():       # a larger batch is refined in pairs (the persistent buffers exist for the path's two batch sizes)
        three = net.forward_log(torch.cat((a, b, a), 0), feats)
    assert three.shape[0] == 3 and torch.equal(three[0:1], ra) and torch.equal(three[1:2], rb) and torch.equal(three[2:3], ra)


@pytest.mark.parametrize("N,H,W,Cin,Cout", [(2, 24, 40, 128, 128), (1, 19, 35, 64, 64)])
def test_rnet_block_in_the_winograd_domain(N, H, W, Cin, Cout):
    """nrgbd_conv_wino_rnet_f32: conv2d_leakyRelu (m_submodule.py:18-27) = 3x3 conv + bias + LeakyReLU(0.01) on the persistent
    Winograd kernel (conv0 / conv0_1 of Refine.py:51-56) vs torch in float64."""
    from neuralrgbd_amd import ops
    x = _rand(N, Cin, H, W, seed=21)
    w = _rand(Cout, Cin, 3, 3, seed=22, scale=0.05)
    b = _rand(Cout, seed=23, scale=0.2)
    want = F.leaky_relu(F.conv2d(x.double(), w.double(), b.double(), 1, 1), 0.01)
    got = ops.conv_wino_rnet(x.permute(0, 2, 3, 1).contiguous(), ops.conv_wino_pack(w), Cout, bias=b, lrelu=True)
    err = (got.permute(0, 3, 1, 2).double() - want).abs().max().item()
    print("[parity] R-Net block (Winograd) N%d %dx%d %d->%d: max|d vs fp64|=%.2e" % (N, H, W, Cin, Cout, err))
    assert err < 2e-5 * max(1.0, want.abs().max().item())


@pytest.mark.parametrize("N,C,H,W,slope", [(1, 128, 16, 24, 0.01), (2, 96, 32, 48, 0.01), (1, 64, 64, 96, 1.0), (1, 256, 9, 11, 0.01), (1, 8, 5, 7, 0.01)])
def test_bias_leaky_relu_fused_forward_and_backward(N, C, H, W, slope):
    """autograd.BiasLeakyReLUCL (the bias + LeakyReLU tail of the R-Net blocks under autograd, m_submodule.py:18-27,36-45; slope 1 =
    the plain bias of Refine.py:71) against torch's add + leaky_relu and their autograd in float64."""
    from neuralrgbd_amd.autograd import BiasLeakyReLUCL
    g = torch.Generator().manual_seed(C + H)
    x = torch.randn(N, C, H, W, generator=g).to(DEV).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    b = torch.randn(C, generator=g).to(DEV).requires_grad_(True)
    gy = torch.randn(N, C, H, W, generator=g).to(DEV)
    y = BiasLeakyReLUCL.apply(x, b, slope)
    gx, gb = torch.autograd.grad(y, (x, b), gy)
    xd, bd = x.detach().double().requires_grad_(True), b.detach().double().requires_grad_(True)
    yd = F.leaky_relu(xd + bd.view(1, -1, 1, 1), slope)
    gxd, gbd = torch.autograd.grad(yd, (xd, bd), gy.double())
    assert (y.double() - yd).abs().max().item() < 1e-6
    assert (gx.double() - gxd).abs().max().item() < 1e-6
    assert (gb.double() - gbd).abs().max().item() < 2e-5 * max(1.0, gbd.abs().max().item())


@pytest.mark.parametrize("N,H,W,Cin,ldx,Cout", [(2, 37, 53, 67, 80, 3), (1, 16, 16, 16, 16, 1), (1, 40, 24, 131, 144, 3), (1, 21, 19, 32, 48, 4)])
def test_conv_with_a_few_output_channels_on_the_vector_alus(N, H, W, Cin, ldx, Cout):
    """csrc/conv_few.hip (the 3 columns beyond the 64-column groups of the R-Net's 67- / 131-wide layer conv2) against F.conv2d in
    float64: ragged tiles, a pixel stride wider than the channels read, output into a column window of a wider buffer."""
    from neuralrgbd_amd import ops
    g = torch.Generator().manual_seed(N * 100 + Cin)
    x = torch.randn(N, H, W, ldx, generator=g)
    cin_p = (Cin + 15) // 16 * 16
    x[..., Cin:] = 0.0
    w = torch.randn(Cout, Cin, 3, 3, generator=g) * 0.1
    b = torch.randn(Cout, generator=g)
    wp = torch.zeros(Cout, cin_p, 3, 3)
    wp[:, :Cin] = w
    few = wp.reshape(Cout, cin_p // 16, 16, 9).permute(1, 3, 0, 2).contiguous().to(DEV)
    out = torch.full((N, H, W, 64 + 16), 5.0, device=DEV)
    ops.conv2d_few(x.to(DEV), few, bias=b.to(DEV), lrelu=True, out=out, ycoff=64)
    want = F.leaky_relu(F.conv2d(x[..., :Cin].permute(0, 3, 1, 2).double(), w.double(), b.double(), padding=1), 0.01).permute(0, 2, 3, 1)
    err = (out[..., 64:64 + Cout].double().cpu() - want).abs().max().item()
    print("[parity] conv_few %dx%dx%d %d->%d max|d|=%.2e (|y|max %.1f)" % (N, H, W, Cin, Cout, err, want.abs().max().item()))
    assert err < 1e-5 * max(1.0, want.abs().max().item())
    assert bool((out[..., :64] == 5.0).all()) and bool((out[..., 64 + Cout:] == 5.0).all())
    plain = ops.conv2d_few(x.to(DEV), few, bias=None, lrelu=False)
    want2 = F.conv2d(x[..., :Cin].permute(0, 3, 1, 2).double(), w.double(), None, padding=1).permute(0, 2, 3, 1)
    assert (plain.double().cpu() - want2).abs().max().item() < 1e-5 * max(1.0, want2.abs().max().item())
