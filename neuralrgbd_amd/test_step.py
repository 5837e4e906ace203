"""One inference step of the streaming depth filter: KVNET.forward + PREDICT.

Same signature and return values as code/test_utils/test_KVNet.py:19-67 (`test`).  The PREDICT
resample and its clamp are one HIP launch (ops.dpv_resample) instead of a host-built point grid,
an H2D copy and two syncs (homography.py:673-690).
"""
import math

import numpy as np
import torch

from . import homography as warp_homo
from . import ops


def test(model_KV, d_candi, Cam_Intrinsics, t_win_r, Ref_Dats, Src_Dats, Src_CamPoses, BV_predict,
         cam_pose_next=None, R_net=False, Cam_Intrinsics_imgsize=None, ref_indx=None, dpv_valid=None):
    """Returns (dpv, BVs_predict): dpv = R-Net output if R_net else the 1/4-res DPV; BVs_predict [N,D,h,w]."""
    nGPU = 1
    BatchIdx_range = torch.arange(nGPU, dtype=torch.float32)
    ref_frame = torch.cat(tuple(ref_dat['img'].cuda() for ref_dat in Ref_Dats), dim=0)
    src_frames = torch.cat(tuple(
        torch.cat(tuple(f['img'].cuda() for f in traj), dim=0).unsqueeze(0) for traj in Src_Dats), dim=0)
    Src_CamPoses = Src_CamPoses.cuda()

    with torch.no_grad():
        kwargs = {} if dpv_valid is None else {'dpv_valid': dpv_valid}
        dmap_cur_refined, dmap_refined, d_dpv, kv_dpv = model_KV(
            ref_frame=ref_frame, src_frames=src_frames, src_cam_poses=Src_CamPoses,
            BatchIdx=BatchIdx_range, cam_intrinsics=Cam_Intrinsics, BV_predict=BV_predict, **kwargs)
        if BV_predict is None:
            kv_dpv, dmap_refined = d_dpv, dmap_cur_refined

        pad = math.log(1. / float(len(d_candi)))
        BVs_predict = []
        for ibatch in range(d_dpv.shape[0]):
            pose = Src_CamPoses[ibatch, t_win_r] if cam_pose_next is None else cam_pose_next.cuda()
            # the reference's `.inverse()` (:50) in the path's own fixed operation order (nrgbd_pose_inverse): the host
            # LAPACK / rocSOLVER order is opaque and one ulp of rel_Rt moves every resampling point of the DPV
            rel_Rt = ops.pose_inverse(pose.to(dtype=torch.float32).contiguous())
            BVs_predict.append(warp_homo.resample_vol_cuda(
                src_vol=kv_dpv[ibatch].unsqueeze(0), rel_extM=rel_Rt, cam_intrinsic=Cam_Intrinsics[ibatch],
                d_candi=d_candi, padding_value=pad, clamp=(-1000., 0.)).unsqueeze(0))
        BVs_predict = torch.cat(BVs_predict, dim=0)

    return (dmap_refined, BVs_predict) if R_net else (kv_dpv, BVs_predict)


test.__test__ = False   # the reference's function name; not a pytest test
