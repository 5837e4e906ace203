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
    the fp32 matrix cores: forward = csrc/wino_pc.hip (64 -> 64 layers, Winograd domain) / csrc/conv3d.hip; data gradient = the
    same kernel on the output gradient with transposed + flipped weights; weight gradient = csrc/conv3d_wgrad.hip.

    x [D,H,W,Cin] (Cin in {16, 64}), w [64,Cin,3,3,3] -> y [D,H,W,64].
    """

    @staticmethod
    def _conv(x, w):
        """y = conv(x, w): the Winograd-domain kernel (wino_pc.hip) for the 64 -> 64 layers, the direct kernel otherwise."""
        if w.shape[0] == 64 and w.shape[1] == 64:
            return ops.conv_wino(x, ops.conv_wino_pack(w), 64, 3, want_stats=False)[0]
        return ops.conv3d(x, ops.conv3d_pack_weights(w.contiguous()), want_stats=False)[0]

    @staticmethod
    def forward(ctx, x, w):
        y = Conv3dCL._conv(x.contiguous(), w)
        ctx.save_for_backward(x, w)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, w = ctx.saved_tensors
        gy = gy.contiguous()
        gx = gw = None
        if ctx.needs_input_grad[1]:
            gw = ops.conv3d_wgrad(x.contiguous(), gy)
        if ctx.needs_input_grad[0]:
            cin = w.shape[1]
            wt = w.transpose(0, 1).flip(2, 3, 4)                      # [Cin, 64, 3,3,3]: correlation with the flipped kernel
            if cin < 64:                                              # the kernel produces 64 outputs: pad, then slice
                wt = torch.cat((wt, wt.new_zeros(64 - cin, 64, 3, 3, 3)), dim=0)
            gx = Conv3dCL._conv(gy, wt.contiguous())
            if cin < 64:
                gx = gx[..., :cin].contiguous()
        return gx, gw
