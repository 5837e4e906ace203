"""Export epilogue of a (refined) DPV — the drop-in for code/test_utils/export_res.py::export_res_img.

The reference builds a [1,D,H,W] tensor of depth values, multiplies, sums, takes a max, goes to the host twice and casts
with numpy; here expected depth, confidence and both uint16 maps leave ONE kernel (nrgbd_export_depth_u16) and only the
two small uint16 images cross PCIe.  File writing is the reference's (PIL), kept out of the kernel path.
"""
import os

import numpy as np
import torch

from . import homography as _homo
from . import ops

_IMAGENET = {'mean': [0.485, 0.456, 0.406], 'std': [0.229, 0.224, 0.225]}


def depth_conf_u16(BV_measure, d_candi, depth_scale=1000, conf_scale=1000):
    """BV_measure [1,D,H,W] (log-prob, CUDA) -> (dmap f32 [H,W], confmap f32 [H,W], depth_u16, conf_u16) on the device:
    dmap = sum_k exp(BV_k) d_k, confmap = exp(max_k BV_k), u16 = (map * scale).astype(uint16) (export_res.py:43-75).
    One deliberate difference: the kernel SATURATES map * scale to [0, 65535] where numpy's astype wraps modulo 65536 —
    the files are identical whenever depth * scale < 65536 (65.5 m at the default scale), and beyond that a saturated
    pixel replaces a wrapped (meaningless) one.  exp is the path's exp_rn (correctly rounded, same operation sequence as
    the CPU oracle): the uint16 maps are bit-identical to the oracle's."""
    assert BV_measure.shape[0] == 1 and BV_measure.shape[1] == len(d_candi)
    d_dev = _homo._d_candi_dev(d_candi, BV_measure.device)
    return ops.export_depth_u16(BV_measure[0], d_dev, float(depth_scale), float(conf_scale))


def export2pgm(fpath, im):
    """code/mio/imgIO.py:9-10."""
    import PIL.Image as image
    image.fromarray(im).convert('I').save(fpath)


def export_res_img(ref_dat, BV_measure, d_candi, resfldr, batch_idx, depth_scale=1000, conf_scale=1000):
    """Same signature and files as the reference: img_%05d.png, d_%05d.pgm, conf_%05d.pgm in `resfldr`."""
    _, _, du, cu = depth_conf_u16(BV_measure, d_candi, depth_scale, conf_scale)
    os.makedirs(resfldr, exist_ok=True)
    img = ref_dat['img'].squeeze().cpu().permute(1, 2, 0).numpy()
    img = np.clip(img * np.array(_IMAGENET['std']) + np.array(_IMAGENET['mean']), 0, 1)   # export_res.py _un_normalize
    import PIL.Image as image
    image.fromarray((img * 255).astype(np.uint8)).save('%s/img_%05d.png' % (resfldr, batch_idx))
    export2pgm('%s/d_%05d.pgm' % (resfldr, batch_idx), du.cpu().numpy())
    export2pgm('%s/conf_%05d.pgm' % (resfldr, batch_idx), cu.cpu().numpy())
