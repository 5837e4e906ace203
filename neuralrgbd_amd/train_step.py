"""One training iteration of KVNET — mirror of code/train_utils/train_KVNet.py:20-203 (`train`) for the
one-process-per-GPU design.

Differences from the reference, all forced by replacing single-process `DataParallel` (train_KVNet.py:261-262):
  * each process owns ONE trajectory (N = 1: the reference asserts N = 1 per replica anyway, KVNET.py:116);
  * the `loss / nGPU` + DataParallel reduce-add (:149-152) becomes an all-reduce of the gradient (sum / world)
    right after `backward()` — pass `grad_reducer=neuralrgbd_amd.distributed.GradAllReduce(model)`;
  * only loss_type 'NLL' (what every training script of the reference uses; 'L1' needs the DGF net).
Forward/backward run the fused HIP sampling kernels behind autograd (neuralrgbd_amd.autograd); the PREDICT step
at the end is the same single launch as at inference.
"""
import math

import torch
import torch.nn.functional as F

from . import homography as warp_homo
from .misc import depth_val_regression, valid_dpv


def train(nGPU, model_KV, optimizer_KV, t_win_r, d_candi, Ref_Dats, Src_Dats, Src_CamPoses, BVs_predict,
          Cam_Intrinsics, refine_dup=False, weight_var=.001, loss_type='NLL', mGPU=False,
          Cam_Intrinsics_spatial_up=None, return_confmap_up=False, grad_reducer=None):
    """Returns (r_dpv, BVs_predict_out, loss, dmap_kv_lowres, dmap_kv_highres) — the two depth maps are device
    tensors (the reference stacks them with the ground truth into numpy arrays for TensorBoard)."""
    if loss_type != 'NLL' or refine_dup:
        raise NotImplementedError("only the NLL loss without depth up-sampling is on this path")
    if len(Ref_Dats) != 1:
        raise AssertionError("one trajectory per process (N = 1 per replica)")
    dev = next(model_KV.parameters()).device
    ref_frame = torch.cat(tuple(r['img'] for r in Ref_Dats), dim=0).to(dev)
    src_frames = torch.cat(tuple(torch.cat(tuple(f['img'] for f in traj), dim=0).unsqueeze(0) for traj in Src_Dats),
                           dim=0).to(dev)
    poses = Src_CamPoses.to(dev)
    optimizer_KV.zero_grad()

    valid = valid_dpv(BVs_predict) if isinstance(BVs_predict, torch.Tensor) else False
    dmap_cur_refined, dmap_refined, d_dpv, kv_dpv = model_KV(
        ref_frame=ref_frame, src_frames=src_frames, src_cam_poses=poses, BatchIdx=torch.zeros(1),
        cam_intrinsics=Cam_Intrinsics, BV_predict=BVs_predict if valid else None, dpv_valid=True if valid else None)

    # train_KVNet.py:103-120: NLL on the 1/4-res DPV and on its R-Net refinement, for the measurement and the update
    depth_ref = Ref_Dats[0]['dmap'].to(dev)                        # [1,h,w] int64 bin indices, 0 = ignore
    depth_ref_imgsize = Ref_Dats[0]['dmap_imgsize_digit'].to(dev)  # [1,H,W]
    loss = F.nll_loss(d_dpv, depth_ref, ignore_index=0)
    loss = loss + F.nll_loss(dmap_cur_refined, depth_ref_imgsize, ignore_index=0)
    if valid:
        loss = loss + F.nll_loss(kv_dpv, depth_ref, ignore_index=0)
        loss = loss + F.nll_loss(dmap_refined, depth_ref_imgsize, ignore_index=0)

    loss.backward()
    if grad_reducer is not None:
        grad_reducer()                 # RCCL all-reduce (sum / world) of the 21 MB fp32 gradient
    optimizer_KV.step()

    # PREDICT on the detached DPV (train_KVNet.py:155-171)
    with torch.no_grad():
        kv = kv_dpv.detach()
        rel_Rt = torch.linalg.inv(poses[0, t_win_r])
        BVs_predict_out = warp_homo.resample_vol_cuda(
            src_vol=kv, rel_extM=rel_Rt, cam_intrinsic=Cam_Intrinsics[0], d_candi=d_candi,
            padding_value=math.log(1. / float(len(d_candi))), clamp=(-1000., 0.)).unsqueeze(0)
        r_dpv = dmap_cur_refined.detach()
        dmap_kv_lowres = depth_val_regression(kv, d_candi, BV_log=True)
        dmap_kv_highres = depth_val_regression(dmap_refined.detach(), d_candi, BV_log=True)
    return r_dpv, BVs_predict_out, loss.detach(), dmap_kv_lowres, dmap_kv_highres
