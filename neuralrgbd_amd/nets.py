"""Convolutional sub-networks of the depth path (D-Net feature CNN, K-Net, R-Net).

Parameter containers keep the attribute names of the reference so that its checkpoints
(`kvnet_scannet.tar`, `kvnet_kitti.tar`; 459 `module.`-prefixed keys) load
unchanged:
  feature CNN  <- code/models/psm_submodule.py:76-167 (feature_extraction) wrapped by
                  code/models/basic.py:13-51 (feature_extractor)
  K-Net        <- code/models/basic.py:53-139 (KV_NET_BASIC)
  R-Net        <- code/models/Refine.py:24-132 (RefineNet_DPV_upsample)
The modules below only own parameters and the layer graph; tensors must live on the GPU and
the sampling / softmax work around them goes through the HIP library (neuralrgbd_amd.ops).

Behavioural notes that matter for parity: the reference never calls
.eval(), so every BatchNorm normalises with batch statistics at inference; 58 of the 60
BatchNorm2d are built with track_running_stats=False, the two residual-shortcut ones and the
eleven BatchNorm3d keep (and update) running statistics.
"""
import math

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F


# --------------------------------------------------------------------------- builders
def _conv_bn2d(cin, cout, k, stride, pad, dilation, track=False):
    """Conv2d(no bias) -> BatchNorm2d; children '0','1' (psm_submodule.py:10-16)."""
    p = dilation if dilation > 1 else pad
    return nn.Sequential(
        nn.Conv2d(cin, cout, k, stride=stride, padding=p, dilation=dilation, bias=False),
        nn.BatchNorm2d(cout, track_running_stats=track),
    )


def _conv_bn3d(cin, cout):
    """Conv3d 3x3x3 s1 p1 (no bias) -> BatchNorm3d; children '0','1' (psm_submodule.py:19-23)."""
    return nn.Sequential(nn.Conv3d(cin, cout, 3, stride=1, padding=1, bias=False), nn.BatchNorm3d(cout))


def _conv_lrelu(cin, cout):
    """Conv2d 3x3 p1 (bias) -> LeakyReLU(0.01); children '0','1' (m_submodule.py:18-27)."""
    return nn.Sequential(nn.Conv2d(cin, cout, 3, stride=1, padding=1, bias=True), nn.LeakyReLU())


def _deconv_lrelu(cin, cout):
    """ConvTranspose2d k4 s2 p1 (bias) -> LeakyReLU(0.01) (m_submodule.py:36-45)."""
    return nn.Sequential(nn.ConvTranspose2d(cin, cout, 4, stride=2, padding=1, bias=True), nn.LeakyReLU())


def _he_init(module):
    """N(0, sqrt(2/(k..k*out))) for convs, gamma=1 / beta=0 for norms (basic.py:29-43,97-111)."""
    for m in module.modules():
        if isinstance(m, (nn.Conv2d, nn.Conv3d)):
            fan = int(np.prod(m.kernel_size)) * m.out_channels
            m.weight.data.normal_(0, math.sqrt(2.0 / fan))
        elif isinstance(m, (nn.BatchNorm2d, nn.BatchNorm3d)):
            m.weight.data.fill_(1)
            m.bias.data.zero_()



def _no_kernel(what):
    """The inference path runs on the hand-written kernels or not at all: a shape without an instantiation is an error, never
    a silent hand-over to MIOpen / rocBLAS."""
    from ._lib import NrgbdError
    return NrgbdError("no hand-written kernel for " + what)


# --------------------------------------------------------------------------- channels-last matrix-core path (shared)
class _Act(object):
    """A channels-last activation that is still owed its normalisation: value = act(z*s+t) (+ act(r*s'+t')).
    The next matrix-core conv applies it while loading its input tile (csrc/conv2d.hip, conv3d.hip)."""
    __slots__ = ("z", "ss", "relu", "r", "r_ss", "r_relu")

    def __init__(self, z, ss=None, relu=False, r=None, r_ss=None, r_relu=False):
        self.z, self.ss, self.relu, self.r, self.r_ss, self.r_relu = z, ss, relu, r, r_ss, r_relu


def _packed_weights(owner, conv):
    """B-operand stream of a conv for the matrix-core kernels, re-packed only when its weight changes."""
    from . import ops
    cache = owner.__dict__.setdefault("_wp_cache", {})
    w = conv.weight
    key = (w.data_ptr(), w._version, str(w.device))
    hit = cache.get(id(conv))
    if hit is None or hit[0] != key:
        hit = (key, ops.conv_pack_weights(w.detach().contiguous()))
        cache[id(conv)] = hit
    return hit[1]


def _packed_wino(owner, conv):
    """Winograd-domain weight stream U = G g G^T of a 3x3(x3) conv for csrc/wino_pc.hip, re-packed only when its weight changes."""
    from . import ops
    cache = owner.__dict__.setdefault("_wp_cache", {})
    w = conv.weight
    key = (w.data_ptr(), w._version, str(w.device), "wino")
    hit = cache.get(("wino", id(conv)))
    if hit is None or hit[0] != key:
        hit = (key, ops.conv_wino_pack32(w.detach().contiguous()) if w.shape[0] == 32 and w.dim() == 4 else ops.conv_wino_pack(w.detach().contiguous()))
        cache[("wino", id(conv))] = hit
    return hit[1]


def _packed_wino_dw(owner, conv, mult=1.0):
    """Weight stream of csrc/wino_dw.hip (Winograd along depth as well): U_t = sum_kd G[t][kd] (G g_kd G^T), of mult * w
    (mult = 2^k: the clamped-FMA form of the kernel, whose input arrives scaled by 2^-k)."""
    from . import ops
    cache = owner.__dict__.setdefault("_wp_cache", {})
    w = conv.weight
    key = (w.data_ptr(), w._version, str(w.device), "wino_dw", float(mult))
    hit = cache.get(("wino_dw", id(conv)))
    if hit is None or hit[0] != key:
        wc = w.detach().contiguous()
        hit = (key, ops.conv_wino_dw_pack(wc if mult == 1.0 else wc * float(mult)))
        cache[("wino_dw", id(conv))] = hit
    return hit[1]


def _packed_wino_dw4(owner, conv, mult=1.0):
    """Weight stream of csrc/wino_dw4.hip (F(4,3) along depth on top of the in-plane Winograd form), of mult * w."""
    from . import ops
    cache = owner.__dict__.setdefault("_wp_cache", {})
    w = conv.weight
    key = (w.data_ptr(), w._version, str(w.device), "wino_dw4", float(mult))
    hit = cache.get(("wino_dw4", id(conv)))
    if hit is None or hit[0] != key:
        wc = w.detach().contiguous()
        hit = (key, ops.conv_wino_dw4_pack(wc if mult == 1.0 else wc * float(mult)))
        cache[("wino_dw4", id(conv))] = hit
    return hit[1]


def _relu_unit(owner, bn, count):
    """x_unit = 2^-k of the clamped-FMA convolution forms for an input relu(bn(y)), bn with batch statistics over `count` values
    (ops.relu_unit); 0 when the bound does not apply (running statistics).  Cached per (affine parameters, count): reading
    the parameters synchronises with the host once."""
    from . import ops
    if not (bn.training or not bn.track_running_stats) or not bn.affine:
        return 0.0
    cache = owner.__dict__.setdefault("_wp_cache", {})
    key = (bn.weight.data_ptr(), bn.weight._version, bn.bias.data_ptr(), bn.bias._version, int(count))
    hit = cache.get(("unit", id(bn)))
    if hit is None or hit[0] != key:
        hit = (key, ops.relu_unit(bn.weight, bn.bias, count))
        cache[("unit", id(bn))] = hit
    return hit[1]


def _packed_s2(owner, conv):
    """Weight stream of a stride-2 3x3 conv in its space-to-depth form (ops.conv_s2_pack), re-packed only when the weight changes."""
    from . import ops
    cache = owner.__dict__.setdefault("_wp_cache", {})
    w = conv.weight
    key = (w.data_ptr(), w._version, str(w.device), "s2")
    hit = cache.get(("s2", id(conv)))
    if hit is None or hit[0] != key:
        hit = (key, ops.conv_s2_pack(w.detach().contiguous()))
        cache[("s2", id(conv))] = hit
    return hit[1]


def invalidate_packed_weights(module):
    """Drop every cached packed-weight stream below `module`.  The caches are keyed on (data_ptr, tensor version), which an
    update through `.data` (`w.data.copy_()`, EMA, manual surgery) does not bump — call this after such an update.
    `load_state_dict` and `_apply` (.to / .cuda / .float) of the networks below call it themselves."""
    for m in module.modules():
        m.__dict__.pop("_wp_cache", None)
        m.__dict__.pop("_pk_cache", None)


class _PackedWeightsMixin(object):
    """nn.Module mixin: invalidate the packed-weight caches whenever the parameters are replaced wholesale."""

    def load_state_dict(self, *a, **k):
        out = super().load_state_dict(*a, **k)
        invalidate_packed_weights(self)
        return out

    def _apply(self, fn, *a, **k):
        out = super()._apply(fn, *a, **k)
        invalidate_packed_weights(self)
        return out


# ------------------------------------------------------------------ status word (variance collapse; include/nrgbd.h)
_status_words = {}


def status_word(device):
    """The int32 device word the BatchNorm finalisers count variance-collapsed channels into (csrc/common.hpp
    bn_finalize_channel): one per device, allocated once (its address is baked into captured hipGraphs)."""
    key = str(device)
    w = _status_words.get(key)
    if w is None:
        w = torch.zeros(1, dtype=torch.int32, device=device)
        _status_words[key] = w
    return w


def check_status(device):
    """Raise NrgbdError when a BatchNorm finaliser reported a variance collapse since the last check (one device -> host read of
    4 bytes: called where the path synchronises anyway — misc.valid_dpv, DepthStream's deferred probe).  The reference's two-pass
    statistics would still normalise such a channel; this path cannot (E[y^2] - mean^2 has no digit left), and says so instead of
    handing out a wrong depth map."""
    w = _status_words.get(str(device))
    if w is None:
        return
    # read through a device-side copy, never by a memcpy FROM the word itself: on this stack (ROCm 7.0 / torch 2.10) a direct
    # device -> host copy of this long-lived 4-byte allocation between two replays of an unrelated hipGraph (train_step.TrainGraph)
    # reproducibly changed what the graph's NEXT replay computed (tests/test_gpu_train.py::test_split_train_graph_with_accumulation_
    # equals_eager_accumulation: 8 of 8 runs; reading a clone, or any freshly allocated tensor, never did).  Unexplained — the word is
    # only ever written by a finaliser that found a collapsed channel — and avoided: the clone is one tiny kernel.
    n = int(w.clone().item())
    if n:
        w.zero_()
        from ._lib import NrgbdError
        raise NrgbdError("BatchNorm batch statistics collapsed in %d channel(s) (std / |mean| < 3.2e-3: the variance E[y^2] - mean^2 of "
                         "the convolution epilogues has no correct digit left); the frame's outputs are invalid" % n)


def _bn_scale_shift(bn, stats, count, cm=False):
    """(scale, shift) [C,2] of a BatchNorm: batch statistics from the conv epilogue's partials in train mode (the
    reference never leaves it) incl. the running-statistics side effect; running statistics in eval mode.
    cm: the partials are column-major [2C, rows] (the Winograd kernel's), not [workgroups, 2C]."""
    from . import ops
    if bn.training or not bn.track_running_stats:
        upd = bn.training and bn.track_running_stats
        nbt = None
        if upd and bn.momentum is None:           # cumulative average: the factor needs the counter's value on the host
            bn.num_batches_tracked += 1
        elif upd:
            nbt = bn.num_batches_tracked          # incremented by the finaliser itself (an ATen launch less per layer)
        momentum = bn.momentum if bn.momentum is not None else 1.0 / float(bn.num_batches_tracked)
        fin = ops.bn_finalize_cm if cm else ops.bn_finalize
        return fin(stats, count, bn.weight.detach(), bn.bias.detach(), bn.eps, momentum,
                   bn.running_mean if upd else None, bn.running_var if upd else None, status=status_word(stats.device),
                   batches_tracked=nbt)
    sc = bn.weight.detach() * torch.rsqrt(bn.running_var + bn.eps)
    return torch.stack((sc, bn.bias.detach() - bn.running_mean * sc), dim=1).contiguous()


def _needs_stats(bn):
    return bn.training or not bn.track_running_stats

# --------------------------------------------------------------------------- 2-D feature CNN
def _fused_ok(x):
    """The hand-written BatchNorm path: inference on the GPU (autograd needs the torch modules)."""
    return x.is_cuda and not torch.is_grad_enabled() and (x.shape[2] * x.shape[3]) % 4 == 0


def _conv_bn_act(x, seq, relu, residual=None):
    """conv -> BatchNorm2d(batch stats) -> [ReLU] -> [+ residual]; `seq` = Sequential(conv, bn): the module (autograd) form of a
    trunk layer.  On the GPU every direction runs on the hand-written kernels (autograd.conv2d_module, csrc/bn_train.hip),
    with or without a graph being recorded; on the CPU it is the torch modules (host-side structure tests)."""
    conv, bn = seq[0], seq[1]
    if x.is_cuda and x.dtype == torch.float32:
        from .autograd import conv2d_module, batch_norm_act_cl
        y = conv2d_module(conv, x)           # forward / data gradient / weight gradient on the hand-written kernels
        # BatchNorm2d + ReLU + add in both directions on csrc/bn_train.hip, on the NHWC view of the channels-last map
        y = y.contiguous(memory_format=torch.channels_last)
        r = None if residual is None else residual.contiguous(memory_format=torch.channels_last).permute(0, 2, 3, 1)
        return batch_norm_act_cl(y.permute(0, 2, 3, 1), bn, relu, r).permute(0, 3, 1, 2)
    y = bn(conv(x))
    if relu:
        y = F.relu(y, inplace=True)
    return y if residual is None else y + residual


class ResBlock2d(nn.Module):
    """Two conv-bn with identity / 1x1 shortcut, no activation after the add (psm_submodule.py:31-50)."""

    def __init__(self, cin, cout, stride, shortcut, pad, dilation):
        super().__init__()
        self.conv1 = nn.Sequential(_conv_bn2d(cin, cout, 3, stride, pad, dilation), nn.ReLU(inplace=True))
        self.conv2 = _conv_bn2d(cout, cout, 3, 1, pad, dilation)
        self.downsample = shortcut
        self.stride = stride

    def forward(self, x):
        skip = x if self.downsample is None else _conv_bn_act(x, self.downsample, relu=False)
        y = _conv_bn_act(x, self.conv1[0], relu=True)
        return _conv_bn_act(y, self.conv2, relu=False, residual=skip)


class PSMFeatures(_PackedWeightsMixin, nn.Module):
    """PSMNet-style pyramid feature CNN: 1/2-res stem, 1/4-res trunk, 4-scale SPP, 1x1 head.

    Output: (layer1 [N,32,H/2,W/2], feat [N,feature_dim,H/4,W/4]) when multi_scale, else feat.
    Layer graph follows psm_submodule.py:90-167 (layer3 dilation 1, layer4 dilation 2, SPP
    windows 64/32/16/8 on the 1/4-res map, bilinear up-sampling with align_corners=True).
    """

    SPP_WINDOWS = (64, 32, 16, 8)

    def __init__(self, feature_dim=32, bn_running_avg=False, multi_scale=False):
        super().__init__()
        self.multi_scale = multi_scale
        t = bn_running_avg
        self.firstconv = nn.Sequential(
            _conv_bn2d(3, 32, 3, 2, 1, 1, t), nn.ReLU(inplace=True),
            _conv_bn2d(32, 32, 3, 1, 1, 1, t), nn.ReLU(inplace=True),
            _conv_bn2d(32, 32, 3, 1, 1, 1, t), nn.ReLU(inplace=True),
        )
        self._width = 32
        self.layer1 = self._stage(32, 3, 1, 1, 1)
        self.layer2 = self._stage(64, 16, 2, 1, 1)
        self.layer3 = self._stage(128, 3, 1, 1, 1)
        self.layer4 = self._stage(128, 3, 1, 1, 2)
        for i, win in enumerate(self.SPP_WINDOWS, start=1):
            setattr(self, "branch%d" % i, nn.Sequential(
                nn.AvgPool2d((win, win), stride=(win, win)),
                _conv_bn2d(128, 32, 1, 1, 0, 1, t), nn.ReLU(inplace=True)))
        self.lastconv = nn.Sequential(
            _conv_bn2d(320, 128, 3, 1, 1, 1, t), nn.ReLU(inplace=True),
            nn.Conv2d(128, feature_dim, 1, padding=0, stride=1, bias=False))

    def _stage(self, width, n_blocks, stride, pad, dilation):
        shortcut = None
        if stride != 1 or self._width != width:
            # the shortcut BatchNorm is the only 2-D norm with running statistics (psm_submodule.py:131)
            shortcut = nn.Sequential(nn.Conv2d(self._width, width, 1, stride=stride, bias=False),
                                     nn.BatchNorm2d(width))
        blocks = [ResBlock2d(self._width, width, stride, shortcut, pad, dilation)]
        self._width = width
        blocks += [ResBlock2d(width, width, 1, None, pad, dilation) for _ in range(n_blocks - 1)]
        return nn.Sequential(*blocks)

    def _spp_pools(self, deep):
        """Average pools of the SPP windows (64, 32, 16, 8).  On the fused path the map is read once by the
        hand-written 8x8 kernel and the coarser windows are pooled from that result (equal-size windows: the
        mean of means is the mean).  Under autograd each window is a crop + reshape + mean: one reduction kernel
        forward and an expand backward, instead of avg_pool2d's one-thread-per-output loop over 64x64 elements
        (0.45 ms per window forward at the ScanNet grid, and as much again backward)."""
        # multiples of 8 suffice: avg_pool2d's floor drops the same ragged border at every level (floor(W/k) windows of k
        # pixels = floor((W/8)/(k/8)) windows of k/8 cells) — the ScanNet grid 64x96 used to miss this path and paid
        # 1.8 ms per frame in avg_pool2d's one-thread-per-output loop
        if _fused_ok(deep) and deep.shape[2] % 8 == 0 and deep.shape[3] % 8 == 0:
            from . import ops
            p8 = ops.avgpool8(deep)
            return {8: p8, 16: F.avg_pool2d(p8, 2), 32: F.avg_pool2d(p8, 4), 64: F.avg_pool2d(p8, 8)}
        if deep.is_cuda and torch.is_grad_enabled():
            N, C, H, W = deep.shape
            out = {}
            for k in self.SPP_WINDOWS:
                ph, pw = H // k, W // k                       # avg_pool2d drops the ragged border (floor)
                out[k] = deep[:, :, :ph * k, :pw * k].reshape(N, C, ph, k, pw, k).mean((3, 5))
            return out
        return {w: F.avg_pool2d(deep, (w, w), stride=(w, w)) for w in self.SPP_WINDOWS}


    def _upsample(self, y, size):
        """Bilinear up-sampling (align_corners=True) of the tiny SPP maps.  Under autograd both directions run on csrc/spp.hip
        (autograd.UpsampleBilinearCL): the backward of F.interpolate scatters every output gradient into a handful of inputs
        with atomics (0.82 ms per branch at the ScanNet grid), the hand-written adjoint sums each input in a fixed order."""
        if y.is_cuda and torch.is_grad_enabled() and y.dtype == torch.float32 and y.shape[1] % 4 == 0 and y.shape[1] <= 256:
            from .autograd import UpsampleBilinearCL
            return UpsampleBilinearCL.apply(y, int(size[0]), int(size[1]))
        return F.interpolate(y, size=size, mode="bilinear", align_corners=True)

    # ------------------------------------------------------------------ matrix-core inference path
    _MFMA_SHAPES = {(32, 1), (64, 1), (128, 1), (128, 2)}   # (Cout, dilation) instantiated in csrc/conv2d.hip

    def _conv_bn_cl(self, seq, a, relu, materialize=False):
        """Sequential(conv, bn) on a pending channels-last activation -> (pending output, materialised input | None).
        3x3 / stride-1 layers run on csrc/wino_pc.hip / conv2d.hip, the stride-2 3x3 layers as 2x2-window convolutions on the
        space-to-depth image (even map sizes: every grid of the path, image sides are multiples of 4).  A layer shape without a
        kernel raises NrgbdError — there is no vendor-library route."""
        from . import ops
        conv, bn = seq[0], seq[1]
        d = conv.dilation[0]
        mfma = (conv.kernel_size == (3, 3) and conv.stride == (1, 1) and conv.padding == (d, d)
                and conv.in_channels % 16 == 0 and (conv.out_channels, d) in self._MFMA_SHAPES)
        # Winograd-domain kernel (csrc/wino_pc.hip) for every layer it covers (Cin % 32 == 0, Cout % 64 == 0, and its HALF form
        # for the 32 -> 32 half-resolution layers): 0.116 vs 0.193 ms (64 -> 64), 0.37 vs 0.69 ms (128 -> 128), 0.80 vs 1.62 ms
        # (320 -> 128) at config B, and 2.4-3x at the 64x96 grid where 16x16 tiles under-fill the chip; conv2d.hip (direct) serves
        # what is left (16-channel inputs, 1x1 and stride-2 forms).
        wino = (mfma and conv.in_channels % 32 == 0 and ((d in (1, 2) and conv.out_channels % 64 == 0) or (d == 1 and conv.out_channels == 32))
                and ops.conv_wino_supported(a.z.shape[0], a.z.shape[1], a.z.shape[2], conv.in_channels, conv.out_channels, 1))
        if wino:
            z, st, mat = ops.conv_wino(a.z, _packed_wino(self, conv), conv.out_channels, 1, d, x_ss=a.ss, x_relu=a.relu,
                                       res=a.r, res_ss=a.r_ss, res_relu=a.r_relu, materialize=materialize,
                                       want_stats=_needs_stats(bn))
        elif mfma:
            z, st, mat = ops.conv2d(a.z, _packed_weights(self, conv), conv.out_channels, d, x_ss=a.ss, x_relu=a.relu,
                                    res=a.r, res_ss=a.r_ss, res_relu=a.r_relu, materialize=materialize,
                                    want_stats=_needs_stats(bn))
        else:
            mat = a.z if (a.ss is None and a.r is None and not a.relu) else \
                ops.nhwc_act(a.z, a.ss, a.relu, a.r, a.r_ss, a.r_relu)
            s2 = (conv.kernel_size == (3, 3) and conv.stride == (2, 2) and conv.padding == (1, 1) and d == 1
                  and mat.shape[1] % 2 == 0 and mat.shape[2] % 2 == 0 and conv.out_channels in (32, 64))
            if s2:   # stride 2 = a 2x2-window convolution on the space-to-depth image of the input
                z, st = ops.conv2d_taps(ops.space_to_depth2(mat), _packed_s2(self, conv), conv.out_channels, 4,
                                        want_stats=_needs_stats(bn))
            else:
                raise _no_kernel("feature CNN layer %d -> %d, kernel %s, stride %s, dilation %d on a %d x %d map" % (
                    conv.in_channels, conv.out_channels, tuple(conv.kernel_size), tuple(conv.stride), d, mat.shape[1], mat.shape[2]))
        count = z.shape[0] * z.shape[1] * z.shape[2]
        return _Act(z, _bn_scale_shift(bn, st, count, cm=wino), relu), mat

    def _pointwise_bn_cl(self, seq, m, stride=1):
        """1x1 Sequential(conv, bn) on a materialised channels-last tensor: the 1-tap form of csrc/conv2d.hip + statistics."""
        from . import ops
        conv, bn = seq[0], seq[1]
        N, H, W, C = m.shape
        if C % 16 != 0 or conv.out_channels not in (32, 64, 128) or H % stride or W % stride:
            raise _no_kernel("1x1 layer %d -> %d, stride %d on a %d x %d map" % (C, conv.out_channels, stride, H, W))
        # a strided shortcut (psm_submodule.py:127-131) reads every stride-th pixel inside the kernel: no gather pass
        z, st = ops.conv2d_taps(m.contiguous(), _packed_weights(self, conv), conv.out_channels, 1, want_stats=_needs_stats(bn), stride=stride)
        return _Act(z, _bn_scale_shift(bn, st, N * (H // stride) * (W // stride)), False)

    def _block_cl(self, blk, a):
        """BasicBlock (psm_submodule.py:31-50): out = bn(conv2(relu(bn(conv1(x))))) + shortcut(x), no ReLU after the add."""
        y1, m = self._conv_bn_cl(blk.conv1[0], a, relu=True, materialize=True)   # m = the block's input, materialised
        y2, _ = self._conv_bn_cl(blk.conv2, y1, relu=False)
        if blk.downsample is None:
            return _Act(y2.z, y2.ss, False, r=m)
        sk = self._pointwise_bn_cl(blk.downsample, m, blk.downsample[0].stride[0])
        return _Act(y2.z, y2.ss, False, r=sk.z, r_ss=sk.ss)

    def fused_ok(self, x):
        """Inference on the GPU -> the matrix-core trunk (forward_channels_last).  With the 8x16-pixel tiles of the persistent
        Winograd kernel it wins at every SURVEY grid (round 2: trunk 2.6 vs 3.7 ms at 256x384 images, 11.7 vs 19.5 ms at
        1024x768; with round 1's 16x16-tile direct kernel the small grids under-filled the chip and stayed on the vendor
        convolutions)."""
        return x.is_cuda and not torch.is_grad_enabled() and x.dtype == torch.float32

    def forward_channels_last(self, x):
        """Inference on the hand-written kernels: x [N,3,H,W] -> (layer1 [N,H/2,W/2,32], feat [N,H/4,W/4,F]), both
        channels-last.  Same graph as forward() (psm_submodule.py:136-167); every 3x3 stride-1 conv is one fused pass
        (conv on the fp32 matrix cores, BatchNorm statistics in its epilogue, normalise + ReLU + residual in the next
        layer's loader); the stride-2 3x3 convs run as 2x2-window convolutions on the space-to-depth image, the 1x1 convs as
        the 1-tap form of the same kernel, the SPP average pooling on csrc/glue.hip.  Nothing of the frame is a torch op."""
        from . import ops
        conv, bn = self.firstconv[0]
        if x.shape[2] % 4 or x.shape[3] % 4:
            raise _no_kernel("image %d x %d: the plane-sweep grid is the image / 4 (models/basic.py:254-263), both sides must be multiples of 4"
                             % (x.shape[2], x.shape[3]))
        # 3 -> 32, stride 2: the image as a 12(+4)-channel space-to-depth tensor, then a 2x2-window convolution
        z, st = ops.conv2d_taps(ops.space_to_depth2(x.contiguous(), nchw=True), _packed_s2(self, conv), conv.out_channels, 4,
                                want_stats=_needs_stats(bn))
        a = _Act(z, _bn_scale_shift(bn, st, z.numel() // z.shape[-1]), True)
        for i in (2, 4):
            a, _ = self._conv_bn_cl(self.firstconv[i], a, relu=True)
        for blk in self.layer1:
            a = self._block_cl(blk, a)
        half = ops.nhwc_act(a.z, a.ss, a.relu, a.r, a.r_ss, a.r_relu)      # returned to the R-Net and read by layer2
        a = _Act(half)
        for blk in self.layer2:
            a = self._block_cl(blk, a)
        quarter = None
        for k, blk in enumerate(self.layer3):
            y1, m = self._conv_bn_cl(blk.conv1[0], a, relu=True, materialize=True)
            if k == 0:
                quarter = m                                                # layer2's output, needed by the SPP concat
            y2, _ = self._conv_bn_cl(blk.conv2, y1, relu=False)
            if blk.downsample is None:
                a = _Act(y2.z, y2.ss, False, r=m)
            else:
                sk = self._pointwise_bn_cl(blk.downsample, m, blk.downsample[0].stride[0])
                a = _Act(y2.z, y2.ss, False, r=sk.z, r_ss=sk.ss)
        for blk in self.layer4:
            a = self._block_cl(blk, a)
        deep = ops.nhwc_act(a.z, a.ss, a.relu, a.r, a.r_ss, a.r_relu)
        N, h, w, _ = deep.shape
        if h < 64 or w < 64:
            raise _no_kernel("an SPP window of 64 on a %d x %d map (psm_submodule.py:100: images below 256 x 256)" % (h, w))
        if h % 8 == 0 and w % 8 == 0:                                      # see _spp_pools: the floor crops agree at every level
            p8 = ops.avgpool_cl(deep, 8)                                   # csrc/glue.hip: the map is read once, channels-last
            p16 = ops.avgpool_cl(p8, 2)                                    # 2 x 2 of the level below: equal windows, the mean of means
            p32 = ops.avgpool_cl(p16, 2)
            pools = {8: p8, 16: p16, 32: p32, 64: ops.avgpool_cl(p32, 2)}
        else:
            pools = {k: ops.avgpool_cl(deep, k) for k in self.SPP_WINDOWS}
        pyramid, fused = [], []
        for i in (4, 3, 2, 1):
            branch = getattr(self, "branch%d" % i)
            pool = pools[self.SPP_WINDOWS[i - 1]]
            pb = self._pointwise_bn_cl(branch[1], pool)                    # 1x1 conv + BatchNorm statistics on the tiny map
            if pb.ss is not None:
                fused.append((pb.z, pb.ss))                                # normalise + ReLU at the taps of spp_concat
                continue
            y = ops.nhwc_act(pb.z, pb.ss, True).permute(0, 3, 1, 2)        # eval-mode norm: channels-last memory, NCHW view
            y = F.interpolate(y, size=(h, w), mode="bilinear", align_corners=True)
            pyramid.append(y.permute(0, 2, 3, 1))                          # channels-last in memory already
        if len(fused) == 4:
            # BatchNorm + ReLU of the four tiny maps, their bilinear up-sampling and the 320-channel concat in ONE launch
            # (csrc/spp.hip) instead of 4 nhwc_act + 4 upsample_bilinear2d + 3 CatArrayBatchedCopy
            cat = ops.spp_concat(quarter, deep, fused)
        else:
            cat = torch.cat([quarter, deep] + pyramid, dim=3)              # [N,h,w,320]
        y, _ = self._conv_bn_cl(self.lastconv[0], _Act(cat), relu=True)
        head = self.lastconv[2]
        if head.out_channels not in (32, 64, 128):
            raise _no_kernel("feature_dim %d (1x1 head): 32, 64 or 128" % head.out_channels)
        # 1x1 head; BatchNorm + ReLU of lastconv[0] in its loader
        feat, _ = ops.conv2d_taps(y.z, _packed_weights(self, head), head.out_channels, 1, x_ss=y.ss, x_relu=True, want_stats=False)
        return half, feat

    def forward(self, x):
        if x.is_cuda:
            # the whole trunk in channels-last memory, the layout of the hand-written conv / BatchNorm kernels
            x = x.contiguous(memory_format=torch.channels_last)
        stem = x
        for i in (0, 2, 4):
            stem = _conv_bn_act(stem, self.firstconv[i], relu=True)
        half = self.layer1(stem)
        quarter = self.layer2(half)
        deep = self.layer4(self.layer3(quarter))
        size = deep.shape[2:]
        pools = self._spp_pools(deep)
        pyramid = []
        for i in (4, 3, 2, 1):
            branch = getattr(self, "branch%d" % i)          # Sequential(AvgPool2d, conv-bn, ReLU)
            y = _conv_bn_act(pools[self.SPP_WINDOWS[i - 1]], branch[1], relu=True)
            pyramid.append(self._upsample(y, size))
        y = _conv_bn_act(torch.cat([quarter, deep] + pyramid, dim=1), self.lastconv[0], relu=True)
        if y.is_cuda:
            from .autograd import conv2d_module
            feat = conv2d_module(self.lastconv[2], y)       # the 1x1 head on the hand-written kernels in all three directions
        else:
            feat = self.lastconv[2](y)
        return (half, feat) if self.multi_scale else feat


class FeatureExtractor(nn.Module):
    """Wrapper holding the CNN as `.feature_extraction` (basic.py:13-51) — the object shared
    between KVNET.feature_extractor and KVNET.d_net.feature_extraction."""

    def __init__(self, feature_dim=32, bn_running_avg=False, multi_scale=False):
        super().__init__()
        self.feature_extraction = PSMFeatures(feature_dim, bn_running_avg, multi_scale)
        self.multi_scale = multi_scale
        _he_init(self)

    def forward(self, img):
        return self.feature_extraction(img)

    def fused_ok(self, img):
        return self.feature_extraction.fused_ok(img)

    def forward_channels_last(self, img):
        half, feat = self.feature_extraction.forward_channels_last(img)
        return (half, feat) if self.multi_scale else feat


# --------------------------------------------------------------------------- K-Net (3-D)
class KalmanGainNet(_PackedWeightsMixin, nn.Module):
    """K-Net: 12 conv3d 3x3x3 (16->64, 10x 64->64, 64->1), BN3d + ReLU, 4 residual pairs.

    gain = kv_net(volume[1,16,D,h,w]) -> [1,1,D,h,w]   (basic.py:53-139)
    """

    def __init__(self, input_volume_channels, feature_dim=32, if_normalize=False, up_sample_ratio=None):
        super().__init__()
        self.in_channels = input_volume_channels
        self.if_normalize = if_normalize
        self.up_sample_ratio = up_sample_ratio
        f = feature_dim
        self.dres0 = nn.Sequential(_conv_bn3d(input_volume_channels, f), nn.ReLU(), _conv_bn3d(f, f), nn.ReLU())
        for i in (1, 2, 3, 4):
            setattr(self, "dres%d" % i, nn.Sequential(_conv_bn3d(f, f), nn.ReLU(), _conv_bn3d(f, f)))
        self.classify = nn.Sequential(_conv_bn3d(f, f), nn.ReLU(),
                                      nn.Conv3d(f, 1, kernel_size=3, padding=1, stride=1, bias=False))
        _he_init(self)

    # ------------------------------------------------------------------ HIP inference path
    def _layers(self):
        """(conv, bn | None) in execution order: dres0 x2, dres1..4 x2 each, classify x2."""
        seq = [self.dres0[0], self.dres0[2]]
        for i in (1, 2, 3, 4):
            blk = getattr(self, "dres%d" % i)
            seq += [blk[0], blk[2]]
        seq.append(self.classify[0])
        return [(m[0], m[1]) for m in seq] + [(self.classify[2], None)]

    def _packed(self, conv):
        """B-operand stream of a conv, re-packed only when its weight changes."""
        from . import ops
        cache = self.__dict__.setdefault("_wp_cache", {})
        w = conv.weight
        key = (w.data_ptr(), w._version, str(w.device))
        hit = cache.get(id(conv))
        if hit is None or hit[0] != key:
            if conv.out_channels == 1:
                packed = w.detach()[0].reshape(w.shape[1], 27).t().contiguous()  # [27, Cin] tap-major
            else:
                packed = ops.conv3d_pack_weights(w.detach().contiguous())
            hit = (key, packed)
            cache[id(conv)] = hit
        return hit[1]

    def _bn_scale_shift(self, bn, stats, count, cm=False):
        """(scale, shift) of a BatchNorm3d: batch statistics in train mode (the reference never leaves it,
        SURVEY §0.2) incl. the running-statistics side effect; running statistics in eval mode.
        cm: column-major partials [128, tiles] of the Winograd kernel instead of [workgroups, 128]."""
        from . import ops
        use_batch = bn.training or not bn.track_running_stats
        if use_batch:
            upd = bn.training and bn.track_running_stats
            nbt = None
            if upd and bn.momentum is None:
                bn.num_batches_tracked += 1
            elif upd:
                nbt = bn.num_batches_tracked      # incremented by the finaliser itself
            momentum = bn.momentum if bn.momentum is not None else 1.0 / float(bn.num_batches_tracked)
            fin = ops.bn_finalize_cm if cm else ops.bn3d_finalize
            return fin(stats, count, bn.weight.detach(), bn.bias.detach(), bn.eps, momentum,
                       bn.running_mean if upd else None, bn.running_var if upd else None, status=status_word(stats.device),
                       batches_tracked=nbt)
        sc = bn.weight.detach() * torch.rsqrt(bn.running_var + bn.eps)
        return torch.stack((sc, bn.bias.detach() - bn.running_mean * sc), dim=1).contiguous()

    _depth_f43_cin = (16, 64)   # input widths that take the F(4,3) form (the 16 -> 64 first layer: 0.62 -> 0.54 ms at config B)
    _depth_f43 = True           # the 64 -> 64 layers on csrc/wino_dw4.hip where D % 4 == 0 (False: wino_dw.hip, the A/B and the training path's form)
    _split_residual = True      # measured (tools/knet_ab.py, NO_SPLIT=1): K-Net 25.37 -> 24.67 ms at config B when introduced, 22.74 -> 22.40 after the shared strips; identical bits

    def forward_channels_last(self, vol, generation=None):
        """Inference on the hand-written kernels: vol [D,H,W,Cin] (channels-last) -> gain [D,H,W].

        One fused pass per layer: conv on the fp32 matrix cores, BatchNorm statistics in its epilogue, normalise + affine +
        ReLU + residual applied by the next layer's loader.  Same graph as forward() (basic.py:113-132):
            c0 = relu(bn(conv(relu(bn(conv(vol))))))
            c_i = bn(conv(relu(bn(conv(c_{i-1}))))) + c_{i-1}     i = 1..4
            gain = conv(relu(bn(conv(c4))))
        The 3x3x3 layers run on wino_dw.hip (Winograd in all three dimensions: F(2,3) along depth on top of the in-plane
        F(2x2,3x3), 8 multiplies per output voxel) where the grid is whole 8x16 tiles and D is even — every configuration of
        the path —, on wino_pc.hip (12 multiplies) where only that fits, on conv3d.hip (direct, 27) otherwise.
        generation: None = that choice; "wino_pc" / "direct" = start the choice at that kernel (tests compare the kernels on
        the whole stack; nothing in the package passes it)."""
        from . import ops
        if self.if_normalize or self.up_sample_ratio is not None:
            raise NotImplementedError("if_normalize / up_sample_ratio are never enabled by the reference scripts")
        if generation not in (None, "wino_pc", "direct"):
            raise ValueError("generation: None | 'wino_pc' | 'direct'")
        D, H, W, C = vol.shape
        if C != self.in_channels:
            raise AssertionError("Input volume should have correct # of channels !")
        L = self._layers()
        count = D * H * W
        need_stats = lambda bn: bn.training or not bn.track_running_stats

        def run(i, x, x_ss, x_relu, res=None, materialize=False):
            conv, bn = L[i]
            cm = False
            # relu(bn(previous output)) as a clamped FMA (wino_dw.hip CLAMP): the previous layer's BatchNorm bounds its output
            unit = _relu_unit(self, L[i - 1][1], count) if (i > 0 and x_ss is not None and x_relu and res is None and not materialize
                                                              and generation is None) else 0.0
            if res is not None and self._split_residual:
                # the residual layers as TWO launches: in = bn(x) + res materialised by one HBM-bound pass (nrgbd_nhwc_act), then the
                # plain form of the convolution on it.  In the fused form the producers of wino_dw.hip load and add the second
                # operand 2-3 times per tile pair — 10 more vector-memory instructions and ~100 more VALU per stage on SIMDs whose
                # issue slots are what bounds the kernel (profiles/r4_wino_design_probe.txt).  res may be a (tensor, scale/shift, relu)
                # triple: c0 = relu(bn(z1)) is never written on its own (lazy_c0 below), the pass applies it while it adds
                r, r_ss, r_relu = res if isinstance(res, tuple) else (res, None, False)
                x = ops.nhwc_act(x, x_ss, x_relu, r, r_ss, r_relu)
                y, ss, _ = run(i, x, None, False)
                return y, ss, x
            if (generation is None and self._depth_f43 and conv.in_channels in self._depth_f43_cin and conv.out_channels == 64 and res is None
                    and ops.conv_wino_dw4_supported(D, H, W, conv.in_channels, 64)):
                # the ten 64 -> 64 layers with F(4,3) along depth (csrc/wino_dw4.hip: 6 instead of 8 multiplies per output voxel).  Its two
                # input forms are x as it is and relu(bn(x)); a layer that also has to KEEP its activated input (dres1.0) materialises it
                # with one HBM-bound pass first, like the residual layers
                if materialize:
                    x = ops.nhwc_act(x, x_ss, x_relu)
                    y, st = ops.conv_wino_dw4(x, _packed_wino_dw4(self, conv), 64, want_stats=need_stats(bn))
                    return y, self._bn_scale_shift(bn, st, count, cm=True), x
                y, st = ops.conv_wino_dw4(x, _packed_wino_dw4(self, conv, 1.0 / unit if unit else 1.0), 64, x_ss=x_ss, x_relu=x_relu,
                                          want_stats=need_stats(bn), x_unit=unit)
                mat, cm = None, True
            elif (generation is None and conv.in_channels in (16, 64) and conv.out_channels == 64
                    and ops.conv_wino_dw_supported(D, H, W, conv.in_channels, 64)):
                y, st, mat = ops.conv_wino_dw(x, _packed_wino_dw(self, conv, 1.0 / unit if unit else 1.0), 64, x_ss=x_ss, x_relu=x_relu,
                                              res=res, materialize=materialize, want_stats=need_stats(bn), x_unit=unit)
                cm = True
            elif generation != "direct" and (conv.in_channels == 64 or (conv.in_channels == 16 and res is None)) \
                    and ops.conv_wino_supported(D, H, W, conv.in_channels, 64, 3):
                # 64 -> 64 (12 stages per tile) and the first layer 16 -> 64 (3 stages: the odd-stage-count instantiation)
                y, st, mat = ops.conv_wino(x, _packed_wino(self, conv), 64, 3, x_ss=x_ss, x_relu=x_relu, res=res,
                                           materialize=materialize, want_stats=need_stats(bn))
                cm = True
            else:
                y, st, mat = ops.conv3d(x, self._packed(conv), x_ss=x_ss, x_relu=x_relu, res=res,
                                        materialize=materialize, want_stats=need_stats(bn))
            return y, self._bn_scale_shift(bn, st, count, cm=cm), mat

        z, ss, _ = run(0, vol, None, False)                       # dres0.0
        z, ss, _ = run(1, z, ss, True)                            # dres0.2   in = relu(bn(z))
        # dres1.0: in = c0 = relu(bn(z)), which is also the first residual.  Where the layer has a clamped-FMA form that costs what the
        # plain one costs (wino_dw4.hip, wino_dw.hip) c0 is not materialised: the layer reads the raw tensor with its (scale, shift), and
        # the first residual pass (dres2.0's input) applies the same FMA + max to it while it adds — the same roundings, one 1.6 GB pass
        # (0.25 ms at config B) less
        lazy_c0 = (generation is None and self._split_residual and L[2][0].in_channels == 64 and L[2][0].out_channels == 64
                   and (ops.conv_wino_dw4_supported(D, H, W, 64, 64) if self._depth_f43 else ops.conv_wino_dw_supported(D, H, W, 64, 64)))
        if lazy_c0:
            skip = (z, ss, True)
            z, ss, _ = run(2, z, ss, True)
        else:
            z, ss, skip = run(2, z, ss, True, materialize=True)   # dres1.0   in = c0 (kept as the residual)
        z, ss, _ = run(3, z, ss, True)                            # dres1.2
        for i in (4, 6, 8):                                       # dres2.0, dres3.0, dres4.0: in = bn(z) + c_{i-1}
            z, ss, skip = run(i, z, ss, False, res=skip, materialize=True)
            z, ss, _ = run(i + 1, z, ss, True)
        z, ss, _ = run(10, z, ss, False, res=skip)                # classify.0: in = c4
        conv, _ = L[11]
        return ops.conv3d_cout1(z, self._packed(conv), x_ss=ss, x_relu=True)  # classify.2

    def forward_channels_last_autograd(self, vol, grad_channel=None):
        """Training path: same graph on channels-last activations with the convolutions (forward, data gradient and
        weight gradient) on the hand-written matrix-core kernels (autograd.Conv3dCL); BatchNorm3d / ReLU / adds are
        ordinary torch autograd ops applied in place of the layout (no NCDHW round trips).  vol [D,H,W,Cin] -> [D,H,W].
        grad_channel: the ONE input channel whose gradient the caller needs (KVNET: the last, BV_cur - BV_predict; the others are
        warped images) — the other channels' gradient is then returned as zeros; None: all of them."""
        from .autograd import Conv3dCL, Conv3dCout1CL, batch_norm_act_cl
        L = self._layers()

        def cbr(x_cl, i, relu, res=None):
            """conv -> BatchNorm3d -> [ReLU] -> [+ res].  [D,H,W,C] viewed as (voxels, C) is exactly the (N, C) form of batch_norm:
            same statistics, same running-statistics update; csrc/bn_train.hip in both directions."""
            conv, bn = L[i]
            return batch_norm_act_cl(Conv3dCL.apply(x_cl, conv.weight, id(conv.weight), grad_channel if i == 0 else None), bn, relu, res)

        x = cbr(vol, 0, True)
        x = cbr(x, 1, True)
        for i in (2, 4, 6, 8):
            x = cbr(cbr(x, i, True), i + 1, False, x)
        y = cbr(x, 10, True)
        # classify.2 = Conv3d(64, 1): a 27-tap stencil per voxel in all three directions (autograd.Conv3dCout1CL; the vendor
        # weight-gradient of this layer alone costs 58 ms at the ScanNet grid, the layer zero-padded to 64 outputs on the matrix-core
        # kernels — rounds 2-5 — 1.0 ms, of which 63/64 multiplies zeros)
        w1 = L[11][0].weight
        if w1.shape[1] == 64 and y.is_cuda:
            return Conv3dCout1CL.apply(y, w1, id(w1))
        w_pad = torch.cat((w1, w1.new_zeros(63, *w1.shape[1:])), dim=0)
        return Conv3dCL.apply(y, w_pad, id(w1))[..., 0]

    def forward(self, volume):
        if volume.shape[1] != self.in_channels:
            raise AssertionError("Input volume should have correct # of channels !")
        x = self.dres0(volume.contiguous())
        for i in (1, 2, 3, 4):
            x = getattr(self, "dres%d" % i)(x) + x
        out = self.classify(x)
        if self.if_normalize:
            out = F.log_softmax(out, dim=2)
        if self.up_sample_ratio is not None:
            d, h, w = volume.shape[2:]
            out = F.interpolate(out, (self.up_sample_ratio * d, h, w), mode="trilinear", align_corners=True)
        return out


# --------------------------------------------------------------------------- R-Net (DPV up-sampler)
class DPVUpsampleNet(_PackedWeightsMixin, nn.Module):
    """R-Net: 2-level conv / transposed-conv decoder over the DPV (D as channels) and the
    1/4-, 1/2- and full-resolution image features; log-softmax over D at the end.

    r_net(dpv[1,D,h,w] (probabilities), [feat 1/4, feat 1/2, image]) -> [1,D,4h,4w] log-prob
    (Refine.py:24-107; transposed convs start as bilinear kernels, :121-132).
    """

    def __init__(self, C0, C1, C2, D=64, upsample_D=False):
        super().__init__()
        cin = D + C0
        D0 = 2 * D if upsample_D else D
        D1 = 2 * D0 if upsample_D else D
        self.conv0 = _conv_lrelu(cin, cin)
        self.conv0_1 = _conv_lrelu(cin, cin)
        self.trans_conv0 = _deconv_lrelu(cin, D0)
        self.conv1 = _conv_lrelu(D0 + C1, D0 + C1)
        self.conv1_1 = _conv_lrelu(D0 + C1, D0 + C1)
        self.trans_conv1 = _deconv_lrelu(D0 + C1, D1)
        self.conv2 = _conv_lrelu(D1 + C2, D1 + C2)
        self.conv2_1 = _conv_lrelu(D1 + C2, D1)
        self.conv2_2 = nn.Conv2d(D1, D1, kernel_size=3, stride=1, padding=1, bias=True)
        self._init()

    def _init(self):
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                fan = m.kernel_size[0] * m.kernel_size[1] * m.out_channels
                m.weight.data.normal_(0, math.sqrt(2.0 / fan))
            elif isinstance(m, nn.ConvTranspose2d):
                n = m.kernel_size[1]
                factor = (n + 1) // 2
                center = factor - 1 if n % 2 == 1 else factor - 0.5
                og = np.ogrid[:n, :n]
                bil = (1 - abs(og[0] - center) / factor) * (1 - abs(og[1] - center) / factor)
                m.weight.data.copy_(torch.from_numpy(bil))

    # ------------------------------------------------------------------ hand-written matrix-core path (inference)
    def _widths(self):
        """(D, Dp, C0, C1, C2) of this net, or None when the kernels do not cover it.  Covered: any D <= 128 depth candidates with
        64 / 32 / 3 image-feature channels (every script of the reference: feature_dim 64) and no candidate up-sampling.  The
        kernels are instantiated for Dp = 64 and 128 candidate channels; another D runs ZERO-PADDED to the next of the two: the
        candidate channels of every buffer are Dp wide (exp(-inf) = 0 in the padding), the weights are embedded with zero rows /
        columns for it (`_embedded`), and the padding of the last layer carries a -1e30 bias so that the log-softmax over the Dp
        channels of a pixel is the log-softmax over its D real ones.  A zero operand adds an exact zero to an fp32 FMA chain, so
        the result is the one a D-wide instantiation with the same summation order would give."""
        D = self.conv2_2.out_channels
        C0 = self.conv0[0].in_channels - D
        C1 = self.conv1[0].in_channels - D
        C2 = self.conv2[0].in_channels - D
        ok = 1 <= D <= 128 and (C0, C1, C2) == (64, 32, 3) and self.trans_conv0[0].out_channels == D \
            and self.trans_conv1[0].out_channels == D and self.conv2_1[0].out_channels == D
        return (D, 64 if D <= 64 else 128, C0, C1, C2) if ok else None

    def mfma_ok(self, dpv):
        """Inference on the GPU -> the hand-written kernels (forward_log), at EVERY grid and candidate count the reference's
        scripts use: the six conv2d_leakyRelu layers and conv2_2 on the Winograd kernel's R-Net form (csrc/wino_pc.hip, 8x16-pixel
        tiles of a persistent launch), the transposed convolutions on csrc/conv2d.hip, the log-softmax on csrc/softmax.hip."""
        return dpv.is_cuda and not torch.is_grad_enabled() and dpv.dtype == torch.float32

    def _embedded(self):
        """{layer: (weight, bias)} in the channel layout of the kernels' buffers: a [candidates (D) | image features (C)] concat
        lives as [D real | Dp - D zero | C], a D-wide output as [D real | Dp - D zero].  D == Dp: the module's own tensors."""
        D, Dp, C0, C1, C2 = self._widths()
        mods = {"conv0": self.conv0[0], "conv0_1": self.conv0_1[0], "trans_conv0": self.trans_conv0[0], "conv1": self.conv1[0],
                "conv1_1": self.conv1_1[0], "trans_conv1": self.trans_conv1[0], "conv2": self.conv2[0], "conv2_1": self.conv2_1[0],
                "conv2_2": self.conv2_2}
        if D == Dp:
            return {k: (m.weight.detach(), m.bias.detach()) for k, m in mods.items()}
        dev = self.conv2_2.weight.device
        cat = lambda C: torch.cat((torch.arange(D, device=dev), Dp + torch.arange(C, device=dev)))      # source channel -> position
        plain = torch.arange(D, device=dev)
        io = {"conv0": (cat(C0), Dp + C0, cat(C0), Dp + C0), "conv0_1": (cat(C0), Dp + C0, cat(C0), Dp + C0),
              "trans_conv0": (cat(C0), Dp + C0, plain, Dp), "conv1": (cat(C1), Dp + C1, cat(C1), Dp + C1),
              "conv1_1": (cat(C1), Dp + C1, cat(C1), Dp + C1), "trans_conv1": (cat(C1), Dp + C1, plain, Dp),
              "conv2": (cat(C2), Dp + C2, cat(C2), Dp + C2), "conv2_1": (cat(C2), Dp + C2, plain, Dp), "conv2_2": (plain, Dp, plain, Dp)}
        out = {}
        for k, m in mods.items():
            in_idx, cin_p, out_idx, cout_p = io[k]
            w, b = m.weight.detach(), m.bias.detach()
            tr = isinstance(m, nn.ConvTranspose2d)            # [Cin, Cout, 4, 4]
            if tr:
                w = w.transpose(0, 1)
            t = w.new_zeros((w.shape[0], cin_p) + tuple(w.shape[2:]))
            t[:, in_idx] = w
            wp = w.new_zeros((cout_p, cin_p) + tuple(w.shape[2:]))
            wp[out_idx] = t
            bp = b.new_full((cout_p,), -1e30 if k == "conv2_2" else 0.0)    # conv2_2: the padding drops out of the log-softmax
            bp[out_idx] = b
            out[k] = ((wp.transpose(0, 1) if tr else wp).contiguous(), bp)
        return out

    def _rnet_packed(self):
        """Per layer a list of output-column slices (packed B-operand stream, padded bias, kernel width, first column,
        valid columns); transposed convs additionally per sub-pixel phase.  Cached (see invalidate_packed_weights)."""
        from . import ops
        cache = self.__dict__.setdefault("_pk_cache", {})
        key = tuple((p.data_ptr(), p._version) for p in self.parameters())
        if cache.get("key") == key:
            return cache["val"]
        pad16 = lambda c: (c + 15) // 16 * 16
        pad32 = pad16                                # pixel width of the concat buffers = whole 16-channel Winograd stages (an odd count has its own instantiation)
        emb = self._embedded()

        def deconv(name):
            w, b = emb[name]                                   # [Cin, Cout, 4, 4]
            res = {}
            for pa in (0, 1):
                for pb in (0, 1):
                    ky = [3, 1] if pa == 0 else [2, 0]        # kernel row that links input row y-1+pa+ty to output row 2y+pa
                    kx = [3, 1] if pb == 0 else [2, 0]
                    wph = w[:, :, ky][:, :, :, kx].permute(1, 0, 2, 3)                # [Cout, Cin, 2, 2]
                    res[(pa, pb)] = [(ops.conv_pack_weights(wph[c0:c0 + 64].contiguous()), b[c0:c0 + 64].contiguous(), 64, c0, 64)
                                     for c0 in range(0, w.shape[1], 64)]
            # all four phases for one launch (mode 3): per 64-column slice, the phases' streams back to back
            res["all"] = [(torch.cat([res[(pa, pb)][i][0] for pa in (0, 1) for pb in (0, 1)]),) + res[(0, 0)][i][1:]
                          for i in range(len(res[(0, 0)]))]
            return res

        def wino(name):
            """Winograd-domain stream of a conv2d_leakyRelu layer for the persistent kernel's R-Net form, widths padded with zero
            weights to Cin % 16 == 0 (the buffer it reads) and Cout % 64 == 0; (stream, bias, packed columns, valid columns)."""
            w, b = emb[name]
            cin_p, cout_p = pad32(w.shape[1]), (w.shape[0] + 63) // 64 * 64
            wp = w.new_zeros(cout_p, cin_p, 3, 3)
            wp[:w.shape[0], :w.shape[1]] = w
            bp = b.new_zeros(cout_p)
            bp[:w.shape[0]] = b
            extra = w.shape[0] % 64
            if 0 < extra <= 4 and w.shape[0] > 64:
                # 67 = 64 + 3 (131 = 128 + 3): the three columns beyond the groups on the vector ALUs (csrc/conv_few.hip: 1,809 FMAs
                # per pixel from an LDS tile) — as a 32-column HALF pass they cost a whole second input transform of the full-resolution
                # buffer (0.38 ms per frame at config B)
                full = w.shape[0] - extra
                few = wp[full:full + extra].reshape(extra, cin_p // 16, 16, 9).permute(1, 3, 0, 2).contiguous()    # [block][tap][co][16]
                return (ops.conv_wino_pack(wp[:full].contiguous()), bp[:full].contiguous(), full, full,
                        ("few", few, bp[full:full + extra].contiguous(), extra, full))
            if 0 < extra <= 32 and w.shape[0] > 64:
                # a few columns beyond a multiple of 64 (67 = 64 + 3, 131 = 128 + 3): the whole groups in one launch, the rest as a
                # 32-column slice on the kernel's HALF form (its stream: the 64-column one with the upper half zero) instead of a
                # 64-column group that is 95 % padding
                full = w.shape[0] - extra
                tail = wp.new_zeros(64, cin_p, 3, 3)
                tail[:extra] = wp[full:full + extra]
                return (ops.conv_wino_pack(wp[:full].contiguous()), bp[:full].contiguous(), full, full,
                        (ops.conv_wino_pack(tail), bp[full:full + 32].contiguous(), 32, extra, full))
            return (ops.conv_wino_pack(wp.contiguous()), bp.contiguous(), cout_p, w.shape[0])

        val = {"t0": deconv("trans_conv0"), "t1": deconv("trans_conv1")}
        for name in ("conv0", "conv0_1", "conv1", "conv1_1", "conv2", "conv2_1", "conv2_2"):
            val[name + "_w"] = wino(name)
        cache["key"], cache["val"] = key, val
        return val

    def _rnet_buffers(self, n, h, w, dev):
        """Persistent channels-last buffers (zero-initialised ONCE: the padding channels of the pixels — 67 -> 80, 131 -> 144,
        and D -> Dp when D is not 64 / 128 — are never written and must stay zero).  One set per batch size (1: first frame,
        2: update)."""
        cache = self.__dict__.setdefault("_buf_cache", {})
        key = (n, h, w, str(dev))
        if key not in cache:
            for k in [k for k in cache if k[1:] != key[1:]]:
                del cache[k]
            D, Dp, C0, C1, C2 = self._widths()
            z = lambda *shape: torch.zeros(shape, dtype=torch.float32, device=dev)
            p32 = lambda c: (c + 15) // 16 * 16                       # whole 16-channel Winograd stages (_rnet_packed pads its weights to the same widths)
            w0, w1, w2 = p32(Dp + C0), p32(Dp + C1), p32(Dp + C2)     # 128, 96, 80 at Dp = 64; 192, 160, 144 at Dp = 128
            cache[key] = {"x0": z(n, h, w, w0), "a0": z(n, h, w, w0), "b0": z(n, h, w, w0),
                          "c1": z(n, 2 * h, 2 * w, w1), "a1": z(n, 2 * h, 2 * w, w1), "b1": z(n, 2 * h, 2 * w, w1),
                          "c2": z(n, 4 * h, 4 * w, w2), "g2": z(n, 4 * h, 4 * w, w2), "h2": z(n, 4 * h, 4 * w, Dp)}
            if D != Dp:     # staging of the candidate planes: rows D.. stay -inf (exp -> the zero padding of the first concat)
                cache[key]["dpv"] = torch.full((n, Dp, h, w), float("-inf"), dtype=torch.float32, device=dev)
        return cache[key]

    def forward_log(self, dpv_log, img_features):
        """Same result as forward(torch.exp(dpv_log), img_features) (models/KVNET.py:128,176 + Refine.py:79-107) with the
        exp fused into the first concat.  Inference on the GPU runs on the hand-written kernels (a net they do not cover —
        candidate up-sampling, other feature widths, D > 128 — raises NrgbdError: there is no vendor-library route);
        `img_features` are ONE image's features, shared by every sample of the batch (KVNET.forward refines BV_cur and DPV of
        a frame as one batch of 2)."""
        first = dpv_log[0] if isinstance(dpv_log, (list, tuple)) else dpv_log
        if not self.mfma_ok(first):
            if isinstance(dpv_log, (list, tuple)):
                dpv_log = torch.cat(list(dpv_log), dim=0)
            return self.forward(torch.exp(dpv_log), img_features)
        from . import ops
        wd = self._widths()
        if wd is None:
            raise _no_kernel("this R-Net (candidate up-sampling, image-feature widths other than 64 / 32 / 3, or more than 128 candidates)")
        D, Dp = wd[0], wd[1]
        quarter, half, full = img_features
        if isinstance(dpv_log, (list, tuple)):     # the frame's two volumes (BV_cur, DPV) as they are: no torch.cat pass
            vols = [v[i:i + 1] for v in dpv_log for i in range(v.shape[0])]
        else:
            vols = [dpv_log[i:i + 1] for i in range(dpv_log.shape[0])]
        n = len(vols)
        _, _, h, w = vols[0].shape
        if n > 2:                                  # buffers are kept for the path's two batch sizes; a larger batch in pairs
            return torch.cat([self.forward_log(vols[i:i + 2], img_features).clone() for i in range(0, n, 2)], dim=0)
        if tuple(quarter.shape) != (1, 64, h, w) or tuple(half.shape) != (1, 32, 2 * h, 2 * w) or tuple(full.shape) != (1, 3, 4 * h, 4 * w):
            raise ValueError("forward_log: features of ONE image at 1/4, 1/2 and full resolution ([1,64,h,w], [1,32,2h,2w], [1,3,4h,4w]) expected")
        dev = first.device
        pk, buf = self._rnet_packed(), self._rnet_buffers(n, h, w, dev)

        def deconv(x, layer, out):
            """ConvTranspose2d(k4, s2, p1) + bias + LeakyReLU: all four sub-pixel phases of every 64-column slice in one launch each."""
            for wp, bias, wdt, c0, valid in layer:
                ops.conv2d_rnet(x, wp, wdt, bias=bias, out=out, ldy=out.shape[-1], ycoff=c0, cout_valid=valid, mode=3)
            return out

        def conv_w(x, name, out=None, lrelu=True):
            """A 3x3 layer on the Winograd kernel's R-Net form (4 instead of 9 multiplies per output and input channel, even after
            padding 96 / 67 outputs to 128): the whole 64-column groups in one launch, a 32-column tail (67 = 64 + 3) on its HALF form."""
            wt = pk[name + "_w"]
            out = ops.conv_wino_rnet(x, wt[0], wt[2], bias=wt[1], lrelu=lrelu, out=out, cout_valid=wt[3])
            if len(wt) > 4:
                t = wt[4]
                if t[0] == "few":
                    ops.conv2d_few(x, t[1], bias=t[2], lrelu=lrelu, out=out, ycoff=t[4])
                else:
                    ops.conv_wino_rnet(x, t[0], t[2], bias=t[1], lrelu=lrelu, out=out, ycoff=t[4], cout_valid=t[3])
            return out

        # level 1/4: cat(exp(dpv), feat) -> conv0 -> conv0_1
        if D != Dp:
            for b in range(n):
                buf["dpv"][b, :D].copy_(vols[b][0])
            vols = [buf["dpv"][b:b + 1] for b in range(n)]
        q_cl = quarter.permute(0, 2, 3, 1)
        x = buf["x0"]
        for b in range(n):
            if q_cl.is_contiguous():
                ops.rnet_pack(vols[b][0].contiguous(), q_cl[0], feat_planar=False, out=x[b:b + 1])
            else:
                ops.rnet_pack(vols[b][0].contiguous(), quarter[0].contiguous(), feat_planar=True, out=x[b:b + 1])
        x = conv_w(conv_w(x, "conv0", buf["a0"]), "conv0_1", buf["b0"])
        # level 1/2: transposed conv (4 sub-pixel phases) straight into channels 0..Dp-1 of the concat buffer; features behind
        c1 = buf["c1"]
        deconv(x, pk["t0"]["all"], c1)
        ops.scatter_channels(half[0], c1, Dp)          # the 1/2-resolution features behind the candidates of every sample (csrc/glue.hip)
        x = conv_w(conv_w(c1, "conv1", buf["a1"]), "conv1_1", buf["b1"])
        # full resolution: Dp + 3 channels in 16-aligned pixels (padding channels zero, with zero weights)
        c2 = buf["c2"]
        deconv(x, pk["t1"]["all"], c2)
        ops.scatter_channels(full[0], c2, Dp)
        x = conv_w(conv_w(c2, "conv2", buf["g2"]), "conv2_1", buf["h2"])
        # conv2_2 + bias on the Winograd kernel (pixels channels-last), then log-softmax over the channels of every pixel in place:
        # the refined DPV is handed out as an [n, D, H, W] VIEW of that channels-last memory (round 4; the direct kernel with the
        # log-softmax epilogue and a planar store took 1.0 ms at config B).  D != Dp: the padding's -1e30 bias keeps it out of the sum
        y = ops.logsoftmax_rows(conv_w(x, "conv2_2", None, lrelu=False)).permute(0, 3, 1, 2)
        return y if D == Dp else y[:, :D]

    def forward(self, dpv_raw, img_features):
        """Refine.py:79-107 as a module call (probabilities in, per-sample features): the autograd-capable composition of the
        hand-written kernels (training; also what a no-grad call on the GPU runs — inference through KVNET uses forward_log).
        On the CPU the same composition falls through to the torch modules (host-side structure tests)."""
        quarter, half, full = img_features
        from .autograd import LogSoftmaxCL, cat_cl, conv2d_module, conv_transpose2d_module, padded_in_width

        def cl(m, x):   # conv2d_leakyRelu block: convolution + fused bias / LeakyReLU(0.01) on the hand-written kernels
            return conv2d_module(m[0], x, act_slope=0.01)

        def tl(m, x):   # conv2dTranspose_leakyRelu block (four sub-pixel phases as one 3x3 launch per direction)
            return conv_transpose2d_module(m[0], x, act_slope=0.01)
        x = cl(self.conv0_1, cl(self.conv0, cat_cl([dpv_raw, quarter], padded_in_width(self.conv0[0]))))
        x = tl(self.trans_conv0, x)
        x = cl(self.conv1_1, cl(self.conv1, cat_cl([x, half], padded_in_width(self.conv1[0]))))
        x = tl(self.trans_conv1, x)
        # conv2 (67 -> 67) runs 96 wide; its padded output (29 exact zeros: zero weights, zero bias, LeakyReLU(0) = 0) feeds
        # conv2_1 as it is — no channel slice, no re-padding at full resolution
        x = conv2d_module(self.conv2[0], cat_cl([x, full], padded_in_width(self.conv2[0])), keep_width=True, act_slope=0.01)
        x = conv2d_module(self.conv2_2, cl(self.conv2_1, x))
        if LogSoftmaxCL.supported(x) and x.permute(0, 2, 3, 1).is_contiguous():
            return LogSoftmaxCL.apply(x)           # channels-last rows kernel; the result stays an NCHW view of that memory
        return torch.log_softmax(x, dim=1)
