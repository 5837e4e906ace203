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
