"""Autograd bindings of the fused sampling kernels (training path, BASELINE config 4).

Inference never touches this module.  Under autograd the D-Net uses
    texels = PackNHWC(features, frames)        backward: channel slice + layout change (no gradient to the images)
    cost   = PlaneSweepCost(texels, ...)       backward: csrc/costvol_bwd.hip
and everything else (log-softmax, K-Net, R-Net, losses) is ordinary torch autograd on the vendor kernels.
"""
import torch

from . import ops


class PackNHWC(torch.autograd.Function):
    @staticmethod
    def forward(ctx, feat, rgb):
        ctx.cf = feat.shape[1]
        return ops.pack_nhwc(feat, rgb)

    @staticmethod
    def backward(ctx, g_tex):
        return g_tex[..., :ctx.cf].permute(0, 3, 1, 2).contiguous(), None


class PlaneSweepCost(torch.autograd.Function):
    """cost [D,h,w] of texels [V+1,h,w,Cp] (last = reference view)."""

    @staticmethod
    def forward(ctx, texels, KR, Kt, rays, d_candi, cx, cy, sigma, C, dist, align_corners):
        V = texels.shape[0] - 1
        cost, _ = ops.costvol(texels[V], texels[:V], KR, Kt, rays, d_candi, cx, cy, sigma, C, dist=dist,
                              align_corners=align_corners, want_cost=True, want_logp=False)
        ctx.save_for_backward(texels, KR, Kt, rays, d_candi)
        ctx.meta = (cx, cy, sigma, C, dist, align_corners)
        return cost

    @staticmethod
    def backward(ctx, g_cost):
        texels, KR, Kt, rays, d_candi = ctx.saved_tensors
        cx, cy, sigma, C, dist, align = ctx.meta
        V = texels.shape[0] - 1
        g_ref, g_src = ops.costvol_bwd(texels[V], texels[:V], KR, Kt, rays, d_candi, cx, cy, sigma, C,
                                       g_cost.contiguous(), dist=dist, align_corners=align)
        return (torch.cat((g_src, g_ref.unsqueeze(0)), dim=0),) + (None,) * 10


class DepthWarp(torch.autograd.Function):
    """Warp of N source images through a per-pixel depth map (warping/homography.py:479-528), differentiable w.r.t. the
    rigid motions (R [N,3,3], t [N,3]) the local bundle adjustment refines (ICP/opt_pose_numerical.py:245-294):
    forward = nrgbd_warp_depth_fwd, backward = nrgbd_warp_depth_bwd.  No gradient is produced for the images or the depth
    map (the reference's caller never asks for one)."""

    @staticmethod
    def forward(ctx, src, dmap, K, R, t, rays):
        R, t = R.contiguous(), t.contiguous()
        ctx.save_for_backward(src, dmap, K, R, t, rays)
        return ops.warp_depth_fwd(src, dmap, K, R, t, rays)

    @staticmethod
    def backward(ctx, g_out):
        src, dmap, K, R, t, rays = ctx.saved_tensors
        g_R, g_t = ops.warp_depth_bwd(src, dmap, K, R, t, rays, g_out.contiguous())
        return None, None, None, g_R, g_t, None


class Conv3dCL(torch.autograd.Function):
    """3x3x3 convolution (stride 1, padding 1, no bias, 64 outputs) on channels-last activations, both directions on
    the fp32 matrix cores: forward = csrc/wino_dw.hip / wino_pc.hip (64 -> 64 layers, Winograd domain) / csrc/conv3d.hip; data gradient = the
    same kernel on the output gradient with transposed + flipped weights; weight gradient = csrc/conv3d_wgrad.hip.

    x [D,H,W,Cin] (Cin in {16, 64}), w [64,Cin,3,3,3] -> y [D,H,W,64].
    """

    @staticmethod
    def _conv(x, w, transposed=False, packed=None):
        """y = conv(x, w) (transposed: with w's data-gradient weights): the Winograd-domain kernels (wino_dw.hip / wino_pc.hip) for
        the 64 -> 64 layers, the direct kernel otherwise.  packed: the weight stream of this call if the caller already has it."""
        if w.shape[0] == 64 and w.shape[1] == 64:
            if ops.conv_wino_dw_supported(x.shape[0], x.shape[1], x.shape[2], 64, 64):   # Winograd along depth too (wino_dw.hip)
                return ops.conv_wino_dw(x, ops.conv_wino_dw_pack(w, transposed) if packed is None else packed, 64, want_stats=False)[0]
            return ops.conv_wino(x, ops.conv_wino_pack(w, transposed) if packed is None else packed, 64, 3, want_stats=False)[0]
        if transposed:
            cin = w.shape[1]
            wt = w.transpose(0, 1).flip(2, 3, 4)                      # [Cin, 64, 3,3,3]: correlation with the flipped kernel
            if cin < 64:                                              # the kernels produce 64 outputs: pad, then slice
                wt = torch.cat((wt, wt.new_zeros(64 - cin, 64, 3, 3, 3)), dim=0)
            if ops.conv_wino_dw_supported(x.shape[0], x.shape[1], x.shape[2], 64, 64):
                # data gradient of the first layer (16 -> 64): a 64 -> 64(16 real) layer in the Winograd domain, 0.31 instead of
                # 0.64 ms on the direct kernel at the training grid
                gx = ops.conv_wino_dw(x, ops.conv_wino_dw_pack(wt.contiguous()), 64, want_stats=False)[0]
            else:
                gx = ops.conv3d(x, ops.conv3d_pack_weights(wt.contiguous()), want_stats=False)[0]
            return gx[..., :cin].contiguous() if cin < 64 else gx
        return ops.conv3d(x, ops.conv3d_pack_weights(w.contiguous()), want_stats=False)[0]

    @staticmethod
    def forward(ctx, x, w):
        x = x.contiguous()
        fwd = bwd = None
        if w.shape[0] == 64 and w.shape[1] == 64 and ctx.needs_input_grad[0]:
            # both weight streams (forward + data gradient) in one launch: the weights changed since the last iteration anyway
            fwd, bwd = ops.conv_wino_pack_both(w, dw=ops.conv_wino_dw_supported(x.shape[0], x.shape[1], x.shape[2], 64, 64))
        y = Conv3dCL._conv(x, w, packed=fwd)
        ctx.save_for_backward(x, w, bwd)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, w, bwd = ctx.saved_tensors
        gy = gy.contiguous()
        gx = gw = None
        if ctx.needs_input_grad[1]:
            gw = ops.conv3d_wgrad(x.contiguous(), gy)
        if ctx.needs_input_grad[0]:
            gx = Conv3dCL._conv(gy, w, transposed=True, packed=bwd)
        return gx, gw


class BatchNormActCL(torch.autograd.Function):
    """y = act(batch_norm(x)) + residual on a channels-last tensor x [..., C] (batch statistics over every leading axis), forward
    and backward on csrc/bn_train.hip — nn.BatchNorm2d / nn.BatchNorm3d + ReLU + residual add of models/psm_submodule.py:10-16,31-50
    and models/basic.py:53-68,71-94 in training.  The running statistics (if any) are updated by the forward kernel."""

    @staticmethod
    def forward(ctx, x, weight, bias, residual, eps, relu, momentum, running_mean, running_var):
        C = x.shape[-1]
        x2 = x.contiguous().view(-1, C)
        r2 = None if residual is None else residual.contiguous().view(-1, C)
        y, coef = ops.bn_cl_fwd(x2, weight, bias, eps, relu, r2, momentum, running_mean, running_var)
        ctx.save_for_backward(x2, coef)
        ctx.relu = relu
        ctx.has_res = residual is not None
        return y.view(x.shape)

    @staticmethod
    def backward(ctx, gy):
        x2, coef = ctx.saved_tensors
        gy2 = gy.contiguous().view(x2.shape)
        gx, gg, gb = ops.bn_cl_bwd(x2, gy2, coef, ctx.relu)
        return gx.view(gy.shape), gg, gb, (gy if ctx.has_res else None), None, None, None, None, None


def batch_norm_act_cl(x_cl, bn, relu, residual=None):
    """Train-mode nn.BatchNorm{2,3}d `bn` (+ ReLU, + residual) applied to the channels-last tensor x_cl [..., C] under autograd.
    NRGBD_TRAIN_BN=vendor keeps torch's batch_norm / relu / add (A/B); shapes bn_train.hip has no form for take that path too."""
    import os
    import torch.nn.functional as F
    C = x_cl.shape[-1]
    use_batch = bn.training or not bn.track_running_stats
    upd = bn.training and bn.track_running_stats
    if upd:
        bn.num_batches_tracked += 1
    m = bn.momentum if bn.momentum is not None else 1.0 / float(bn.num_batches_tracked)
    if (use_batch and x_cl.is_cuda and x_cl.dtype == torch.float32 and bn.affine and ops.bn_cl_supported(x_cl.numel() // C, C)
            and os.environ.get("NRGBD_TRAIN_BN", "native") == "native"):
        if x_cl.numel() // C <= 1:      # nn.BatchNorm's own refusal (torch/nn/functional.py::_verify_batch_size)
            raise ValueError("Expected more than 1 value per channel when training, got input size %s" % (tuple(x_cl.shape),))
        return BatchNormActCL.apply(x_cl, bn.weight, bn.bias, residual, bn.eps, relu, m,
                                    bn.running_mean if upd else None, bn.running_var if upd else None)
    y = F.batch_norm(x_cl.reshape(-1, C), bn.running_mean if bn.track_running_stats else None,
                     bn.running_var if bn.track_running_stats else None, bn.weight, bn.bias, use_batch, m, bn.eps).view_as(x_cl)
    if relu:
        y = torch.relu(y)
    return y if residual is None else y + residual


class Conv2dCL(torch.autograd.Function):
    """3x3 convolution (stride 1, padding = dilation, no bias) of the feature CNN / R-Net under autograd, all three directions
    on the hand-written matrix-core kernels (models/psm_submodule.py:10-16, models/m_submodule.py:18-27 in training,
    train_utils/train_KVNet.py:103-153):
        forward        csrc/wino_pc.hip (Winograd domain; Cin % 32 == 0, Cout % 64 == 0) or csrc/conv2d.hip (direct)
        data gradient  the same kernels on the output gradient with the transposed + flipped weights
        weight grad.   csrc/conv2d_wgrad.hip
    x is an NCHW tensor, ideally in channels_last memory format (then the NHWC views the kernels work on are free); the result
    is an NCHW view of channels-last memory, so a trunk built from this op stays in that format.
    """

    DIRECT = {(32, 1), (64, 1), (96, 1), (128, 1), (128, 2)}    # (Cout, dilation) instantiated in conv2d.hip

    @staticmethod
    def eligible(cin, cout, dil):
        """Forward, data gradient (roles of Cin / Cout swapped) and weight gradient all have a kernel."""
        def fwd(ci, co):
            return (ci % 32 == 0 and co % 64 == 0 and dil in (1, 2)) or (ci % 16 == 0 and (co, dil) in Conv2dCL.DIRECT)
        return fwd(cin, cout) and fwd(cout, cin) and cin % 16 == 0 and cout % 16 == 0

    @staticmethod
    def _conv(x_cl, w, dil, transposed=False, packed=None):
        cout, cin = (w.shape[1], w.shape[0]) if transposed else w.shape[:2]
        if cin % 32 == 0 and cout % 64 == 0:
            return ops.conv_wino(x_cl, ops.conv_wino_pack(w, transposed) if packed is None else packed, cout, 1, dil, want_stats=False)[0]
        if transposed:
            w = w.transpose(0, 1).flip(2, 3)                          # [Cin, Cout, 3, 3]: correlation with the flipped kernel
        return ops.conv2d(x_cl, ops.conv_pack_weights(w.contiguous()), cout, dil, want_stats=False)[0]

    @staticmethod
    def forward(ctx, x, w, dil):
        x_cl = x.permute(0, 2, 3, 1).contiguous()                 # free when x is channels_last
        cout, cin = w.shape[:2]
        fwd = bwd = None
        if cin % 64 == 0 and cout % 64 == 0 and ctx.needs_input_grad[0]:
            fwd, bwd = ops.conv_wino_pack_both(w)                 # both directions run on wino_pc.hip: one packing launch for the two streams
        y = Conv2dCL._conv(x_cl, w, dil, packed=fwd)
        ctx.save_for_backward(x_cl, w, bwd)
        ctx.dil = dil
        return y.permute(0, 3, 1, 2)

    @staticmethod
    def backward(ctx, gy):
        x_cl, w, bwd = ctx.saved_tensors
        gy_cl = gy.permute(0, 2, 3, 1).contiguous()
        gx = gw = None
        if ctx.needs_input_grad[1]:
            gw = ops.conv2d_wgrad(x_cl, gy_cl, ctx.dil)
        if ctx.needs_input_grad[0]:
            gx = Conv2dCL._conv(gy_cl, w, ctx.dil, transposed=True, packed=bwd).permute(0, 3, 1, 2)
        return gx, gw, None


def conv2d_module(conv, x):
    """nn.Conv2d forward for the module (autograd) paths: every 3x3 stride-1 convolution with a kernel in all three directions
    (CUDA, fp32, padding = dilation) goes through Conv2dCL.  NRGBD_TRAIN_CONV=vendor keeps the vendor library for A/B: at the
    64x96 training grid, round 3, the iteration replayed as a hipGraph takes 38.7 ms on the hand-written kernels and 41.4 ms on
    the vendor convolutions (round 2: 53.1 vs 51.4 ms — since then the weight-gradient kernel stages its dY tile in LDS and its
    partials are reduced by 8 slices per workgroup)."""
    import os
    d = conv.dilation[0]
    if (x.is_cuda and x.dtype == torch.float32 and conv.kernel_size == (3, 3) and conv.stride == (1, 1)
            and conv.padding == (d, d) and conv.dilation == (d, d) and conv.groups == 1
            and Conv2dCL.eligible(conv.in_channels, conv.out_channels, d)
            and os.environ.get("NRGBD_TRAIN_CONV", "native") == "native"):
        y = Conv2dCL.apply(x, conv.weight, d)
        return y if conv.bias is None else y + conv.bias.view(1, -1, 1, 1)
    return conv(x)
