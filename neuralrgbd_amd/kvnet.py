"""KVNET — the full D-Net -> R-Net -> K-Net -> DPV-update pipeline behind the reference call surface.

Constructor, `forward` signature, return tuple, sub-module names (`feature_extractor`, `d_net`,
`kv_net`, `r_net`) and state-dict keys follow code/models/KVNET.py:35-39,93-94,185 so that the
callers (test_utils/test_KVNet.py:37-40, train_utils/train_KVNet.py:85-98) and the released
checkpoints work unchanged.  What differs is the execution: the geometry between the
convolutions runs in hand-written gfx950 kernels (neuralrgbd_amd.ops):

    features --pack_nhwc--> texels --costvol (warp+cost+log-softmax, 1 launch)--> BV_cur
    texels[RGB] --warp_volume (warp + K-Net input assembly, 1 launch)--> [1,16,D,h,w]
    K-Net gain + BV_predict --logsoftmax_d--> DPV

so the [D,C,h,w] warped feature tensors, the repeat/transpose/cat temporaries and the per-call
H2D uploads of the reference do not exist.
"""
import numpy as np
import torch
import torch.nn as nn

from . import homography as warp_homo
from . import nets, ops
from .misc import depth_val_regression, valid_dpv


class DNet(nn.Module):
    """D-Net: shared feature CNN -> plane-sweep cost volume -> log-softmax (basic.py:141-323).

    forward(ref_frame [1,3,H,W], src_frames [1,V,3,H,W], src_cam_poses [1,V,4,4]) ->
        BV_cur [1,D,h,w] (log-prob), and with output_features [feat_ref [1,F,h,w], layer1_ref [1,32,H/2,W/2]].
    Also leaves `self.texels` ([V+1,h,w,Cp], last = reference) for the K-Net warp.
    """

    def __init__(self, feature_extraction, cam_intrinsics, d_candi, sigma_soft_max, BV_log=False,
                 normalize=True, use_img_intensity=False, force_img_dw_rate=1, parallel_d=True,
                 output_features=False, refine_costV=False, feat_dist='L2'):
        super().__init__()
        if refine_costV:
            raise NotImplementedError("refine_costV is never enabled by the reference scripts (basic.py:196 is broken)")
        self.feature_extraction = feature_extraction
        self.cam_intrinsics = cam_intrinsics
        self.d_candi = d_candi
        self.sigma_soft_max = sigma_soft_max
        self.BV_log = BV_log
        self.normalize = normalize
        self.use_img_intensity = use_img_intensity
        self.output_features = output_features
        self.feat_dist = feat_dist
        self.align_corners = False  # torch>=1.3 grid_sample default, what the reference runs today
        self.texels = None
        self.rgb4 = None            # [V+1,h,w,4]: the texels' pooled-RGB word as a compact plane (inference; written by pack_nhwc)

    def forward(self, ref_frame, src_frames, src_cam_poses, cam_intrinsics=None, BV_predict=None, debug_ipdb=False):
        assert src_frames.shape[0] == 1, 'dim0 of src_frames should be 0'
        frames = torch.cat((src_frames[0], ref_frame), dim=0)  # batch of V+1: BN statistics span the window
        fe = self.feature_extraction
        # inference: the CNN trunk on the matrix-core kernels, activations channels-last end to end
        feats_cl = hasattr(fe, "forward_channels_last") and fe.fused_ok(frames)
        if feats_cl:
            out = fe.forward_channels_last(frames)
            layer1, feats = out if self.output_features else (None, out)
            V = src_frames.shape[1]
            h, w, F_dim = feats.shape[1:]
        else:
            if self.output_features:
                layer1, feats = fe(frames)
            else:
                layer1, feats = None, fe(frames)
            V = src_frames.shape[1]
            F_dim, h, w = feats.shape[1:]
        C = F_dim + (3 if self.use_img_intensity else 0)
        # avg-pooled RGB appended as channels F..F+2 and the whole window transposed to 16-B texels
        cam = self.cam_intrinsics if cam_intrinsics is None else cam_intrinsics
        dev = feats.device
        K, rays = warp_homo._cam_dev(cam, dev)
        KR, Kt = warp_homo.homography_terms(K, src_cam_poses[0, :, :3, :3], src_cam_poses[0, :, :3, 3])
        cx, cy = cam['intrinsic_M'][0, 2], cam['intrinsic_M'][1, 2]
        d_dev = warp_homo._d_candi_dev(self.d_candi, dev)
        rgb = frames if self.use_img_intensity else None
        if torch.is_grad_enabled() and feats.requires_grad:
            # training: same kernels behind autograd.Function (backward = csrc/costvol_bwd.hip)
            from .autograd import LogSoftmaxD, PackNHWC, PlaneSweepCost
            texels = PackNHWC.apply(feats, rgb)
            cost = PlaneSweepCost.apply(texels, KR, Kt, rays, d_dev, cx, cy, self.sigma_soft_max, C,
                                        self.feat_dist, self.align_corners)
            self.texels, self.rgb4 = texels.detach(), None
            BV = (LogSoftmaxD.apply(cost, None, -1.0) if self.BV_log else torch.softmax(-cost, dim=0)).unsqueeze(0)
        else:
            if rgb is not None:     # + the pooled-RGB word as a compact plane for the K-Net's warp (same launch)
                texels, self.rgb4 = ops.pack_nhwc(feats, rgb, channels_last=feats_cl, want_rgb4=True)
            else:
                texels, self.rgb4 = ops.pack_nhwc(feats, rgb, channels_last=feats_cl), None
            self.texels = texels
            cost, logp = ops.costvol(texels[V], texels[:V], KR, Kt, rays, d_dev, cx, cy, self.sigma_soft_max, C,
                                     dist=self.feat_dist, align_corners=self.align_corners,
                                     want_cost=not self.BV_log, want_logp=self.BV_log)
            BV = logp.unsqueeze(0) if self.BV_log else torch.softmax(-cost.unsqueeze(0), dim=1)

        if BV_predict is not None:  # filtering inside the D-Net (unused by KVNET, basic.py:304-314)
            if not self.BV_log:
                BV = BV * BV_predict
                BV = BV / torch.sum(BV, dim=1).unsqueeze_(1)
            else:
                BV = BV + BV_predict
                if self.normalize:
                    BV = ops.logsoftmax_d(BV[0]).unsqueeze(0)

        if self.output_features:
            if feats_cl:   # NCHW-shaped views of the channels-last tensors (the R-Net's convs take either layout)
                return BV, [feats[V:V + 1].permute(0, 3, 1, 2), layer1[V:V + 1].permute(0, 3, 1, 2)]
            return BV, [feats[V:V + 1], layer1[V:V + 1]]
        return BV


class KVNET(nn.Module):
    """Drop-in for code/models/KVNET.py::KVNET."""

    def __init__(self, feature_dim, cam_intrinsics, d_candi, sigma_soft_max, KVNet_feature_dim,
                 d_upsample_ratio_KV_net, if_refined=True, refineNet_name='DPV', t_win_r=2,
                 refine_channel=3, if_upsample_d=False):
        super().__init__()
        self.t_win_r = t_win_r
        self.feature_dim = feature_dim
        self.KVNet_feature_dim = KVNet_feature_dim
        self.sigma_soft_max = sigma_soft_max
        self.d_upsample_ratio_KV_net = d_upsample_ratio_KV_net
        self.d_candi = d_candi
        self.if_refined = if_refined
        self.refineNet_name = refineNet_name
        self.if_upsample_d = if_upsample_d
        if if_refined and refineNet_name != 'DPV':
            raise NotImplementedError("only the 'DPV' refinement net is on this path (the reference scripts never select 'DGF')")

        dpv_refine = bool(if_refined)
        self.feature_extractor = nets.FeatureExtractor(feature_dim=feature_dim, multi_scale=dpv_refine)
        self.d_net = DNet(self.feature_extractor, cam_intrinsics, d_candi, sigma_soft_max,
                          use_img_intensity=True, BV_log=True, output_features=dpv_refine)
        self.kv_net = nets.KalmanGainNet(3 * (t_win_r * 2 + 1) + 1, feature_dim=KVNet_feature_dim,
                                         up_sample_ratio=d_upsample_ratio_KV_net)
        if if_refined:
            self.r_net = nets.DPVUpsampleNet(int(feature_dim), int(feature_dim / 2), 3,
                                             D=len(d_candi), upsample_D=if_upsample_d)

    def load_state_dict(self, *a, **k):
        out = super().load_state_dict(*a, **k)
        nets.invalidate_packed_weights(self)      # packed B-operand streams are keyed on tensor versions (nets.py)
        return out

    def _refine(self, dpv_log, features):
        if hasattr(self.r_net, "forward_log") and not torch.is_grad_enabled():
            return self.r_net.forward_log(dpv_log, features)      # exp fused into the first concat; matrix-core convs
        return self.r_net(torch.exp(dpv_log), img_features=features)

    # ------------------------------------------------------------------ the frame in two halves (streaming.DepthStream pipelines them)
    def measure(self, ref_frame, src_frames, src_cam_poses):
        """FRONT half of an inference frame: everything that does not depend on the filter state — the feature CNN of the window,
        the texel pack, the fused warp + cost volume + log-softmax (KVNET.py:120-123: `self.d_net(...)`).  Returns the record
        `forward(..., measured=record)` continues from: the D-Net of frame t+1 can run while the K-Net of frame t is still busy
        (a2-a4 of the next frame are independent of this frame's state)."""
        if self.if_refined:
            BV_cur, features = self.d_net(ref_frame, src_frames, src_cam_poses, BV_predict=None)
            features.append(ref_frame)
        else:
            BV_cur, features = self.d_net(ref_frame, src_frames, src_cam_poses, BV_predict=None), None
        return {"BV_cur": BV_cur, "features": features, "texels": self.d_net.texels, "rgb4": self.d_net.rgb4}

    def forward(self, ref_frame, src_frames, src_cam_poses, BatchIdx, cam_intrinsics=None,
                BV_predict=None, mGPU=False, IntMs=None, unit_ray_Ms_2D=None, dpv_valid=None, measured=None):
        """
        ref_frame [1,3,H,W], src_frames [1,V,3,H,W], src_cam_poses [1,V,4,4], BatchIdx [1],
        cam_intrinsics: list of dicts, BV_predict [1,D,h,w] or None.
        Returns (dmap_cur_refined, dmap_refined, BV_cur, DPV); the first / invalid frame returns the
        D-Net pair twice; -1 sentinels when if_refined is False (KVNET.py:136-143,182).
        `dpv_valid` (extension): host-side validity of BV_predict; None = probe the tensor like
        the reference's valid_dpv (one device->host read).
        `measured` (extension): the record of `measure()` for this window — the D-Net is then not run again.
        """
        has_pred = isinstance(BV_predict, torch.Tensor)
        if has_pred and dpv_valid is None:
            dpv_valid = valid_dpv(BV_predict)
        if has_pred and dpv_valid:
            assert BV_predict.shape[0] == 1

        # update branch at inference: the two R-Net calls of the frame (on BV_cur and on DPV; Refine.py is a per-sample
        # network without BatchNorm) run as ONE batch of 2 after the K-Net — twice the workgroups per launch for the
        # quarter-resolution layers, which alone do not fill the chip
        batch_refine = (self.if_refined and has_pred and bool(dpv_valid) and not torch.is_grad_enabled()
                        and hasattr(self.r_net, "forward_log") and self.r_net.mfma_ok(BV_predict))
        if measured is None:
            measured = self.measure(ref_frame, src_frames, src_cam_poses)
        BV_cur, features = measured["BV_cur"], measured["features"]
        if self.if_refined:
            dmap_cur_refined = None if batch_refine else self._refine(BV_cur, features)
        else:
            dmap_cur_refined = -1

        if not has_pred or not dpv_valid:
            return dmap_cur_refined, dmap_cur_refined, BV_cur, BV_cur

        # ---- K-Net: warp the 1/4-res RGB of the sources to the reference for every candidate ----
        texels = measured["texels"]           # [V+1,h,w,Cp]; channels F..F+2 are the pooled RGB
        V = src_frames.shape[1]
        h, w, Cp = texels.shape[1:]
        F_dim = self.feature_dim
        dev = texels.device
        if mGPU:
            K = IntMs.squeeze(0).to(torch.float32)
            rays = unit_ray_Ms_2D.squeeze(0).to(torch.float32).contiguous()
            cx, cy = float(K[0, 2]), float(K[1, 2])
        else:
            cam = cam_intrinsics[int(BatchIdx)]
            K, rays = warp_homo._cam_dev(cam, dev)
            cx, cy = cam['intrinsic_M'][0, 2], cam['intrinsic_M'][1, 2]
        KR, Kt = warp_homo.homography_terms(K, src_cam_poses[0, :, :3, :3], src_cam_poses[0, :, :3, 3])
        # the RGB word of every texel as a compact [V+1,h,w,4] plane: at the texel tensor's 272-B stride every lane of the
        # warp kernel's gathers touched its own cache line; at 16 B per texel four neighbouring taps share one
        rgb4 = measured["rgb4"] if (measured["rgb4"] is not None and texels.shape[-1] == F_dim + 4) else texels[..., F_dim:].contiguous()
        rgb_src, rgb_ref, Cp = rgb4[:V], rgb4[V], rgb4.shape[-1]
        fused = (not torch.is_grad_enabled()) and self.kv_net.in_channels == 16 and self.KVNet_feature_dim == 64 \
            and not self.kv_net.if_normalize and self.kv_net.up_sample_ratio is None
        warp_args = (rgb_src, (h * w * Cp, 1, w * Cp, Cp), rgb_ref, (1, w * Cp, Cp), KR, Kt, rays,
                     warp_homo._d_candi_dev(self.d_candi, dev), cx, cy, V, 3, h, w)
        if fused:   # inference: hand-written MFMA conv3d stack on the channels-last volume
            volume = ops.warp_volume(*warp_args, bv_cur=BV_cur[0], bv_pred=BV_predict[0],
                                     align_corners=self.d_net.align_corners, channels_last=True)
            gain = self.kv_net.forward_channels_last(volume)                # [D,h,w]
            DPV = ops.logsoftmax_d(gain, BV_predict[0]).unsqueeze(0)        # UPDATE
        else:       # autograd path (training): the warped RGB is constant, BV_cur - BV_predict carries the gradient
            warped = ops.warp_volume(*warp_args, align_corners=self.d_net.align_corners)   # [15,D,h,w]
            volume = torch.cat((warped, BV_cur - BV_predict), dim=0)                       # [16,D,h,w]
            if self.kv_net.in_channels == 16 and self.KVNet_feature_dim == 64 and torch.is_grad_enabled():
                gain = self.kv_net.forward_channels_last_autograd(volume.permute(1, 2, 3, 0).contiguous(),
                                                                  grad_channel=volume.shape[0] - 1).unsqueeze(0)   # only BV_cur - BV_predict
            elif volume.is_cuda:
                # a K-Net the kernels have no form for (KVNet_feature_dim != 64, a window other than 5 frames, depth up-sampling:
                # no script of the reference selects one): an error, never a silent hand-over to MIOpen
                raise nets._no_kernel("this K-Net (%d input channels, feature width %d, if_normalize %s, up_sample_ratio %s)" % (
                    self.kv_net.in_channels, self.KVNet_feature_dim, self.kv_net.if_normalize, self.kv_net.up_sample_ratio))
            else:
                gain = torch.squeeze(self.kv_net(volume.unsqueeze(0)), dim=1)   # torch modules on the host (structure tests)
            if gain.is_cuda and gain.dtype == torch.float32:
                from .autograd import LogSoftmaxD
                DPV = LogSoftmaxD.apply(gain, BV_predict, 1.0)                  # UPDATE, softmax.hip in both directions
            else:
                DPV = torch.log_softmax(gain + BV_predict, dim=1)

        if batch_refine:
            both = self.r_net.forward_log((BV_cur, DPV), features)                       # [2,D,H,W]; no concatenation pass
            dmap_cur_refined, dmap_refined = both[0:1], both[1:2]
        else:
            dmap_refined = self._refine(DPV, features) if self.if_refined else -1
        return dmap_cur_refined, dmap_refined, BV_cur, DPV
