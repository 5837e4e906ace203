"""Small helpers of the depth path (the used part of code/mutils/misc.py)."""
import torch

from . import homography as _homo
from . import ops


def valid_dpv(dpv_in):
    """True unless the volume is None or flagged invalid by a NaN at its first element
    (mutils/misc.py:100-115; the NaN convention comes from mdataloader/batch_loader.py:30-43).
    Reads one element back to the host, exactly like the reference."""
    if dpv_in is None:
        return False
    assert isinstance(dpv_in, torch.Tensor), 'input should a Tensor'
    if not 2 <= dpv_in.dim() <= 5:
        raise Exception('wrong dimension for input dpv !')
    ok = not bool(torch.isnan(dpv_in[(0,) * dpv_in.dim()]))
    if dpv_in.is_cuda:
        # the path's own failure report rides on this synchronisation: a BatchNorm whose batch statistics collapsed in the frame
        # that produced dpv_in raises here (nets.check_status) instead of passing on a wrong volume
        from .nets import check_status
        check_status(dpv_in.device)
    return ok


def depth_val_regression(BV_measure, d_candi_cur, BV_log=True):
    """Expected depth sum_d p_d * d of a [1,D,h,w] volume -> [1,h,w] (mutils/misc.py:532-548).
    One kernel instead of a Python loop over D."""
    assert len(d_candi_cur) == BV_measure.shape[1], \
        'BV_measure should have the same # of slices as len(d_candi_cur) !'
    d_dev = _homo._d_candi_dev(d_candi_cur, BV_measure.device)
    logp = BV_measure[0] if BV_log else torch.log(BV_measure[0])
    depth, _ = ops.depth_regress(logp, d_dev, want_conf=False)
    return depth.unsqueeze(0)


def dpv_confidence(BV_measure):
    """max_d log-prob -> [1,h,w] (test_utils/export_res.py:58-59)."""
    d_dev = torch.zeros(BV_measure.shape[1], dtype=torch.float32, device=BV_measure.device)
    _, conf = ops.depth_regress(BV_measure[0], d_dev, want_conf=True)
    return conf.unsqueeze(0)


def split_frame_list(frame_list, t_win_r):
    """ref = frame_list[t_win_r], src = the others in order (mutils/misc.py:509-517)."""
    ref = frame_list[t_win_r]
    src = [f for i, f in enumerate(frame_list) if i != t_win_r]
    return ref, src


def get_entries_list_dict(list_dict, keyname):
    return [d[keyname] for d in list_dict]
