"""ctypes loader of libnrgbd_hip.so — the C-ABI declared in include/nrgbd.h.

There is deliberately NO fallback: if the HIP library is missing or does not export every
symbol of the header, importing the product path raises.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libnrgbd_hip.so")

_c = ctypes
_P = _c.c_void_p
_F = _c.c_float
_I = _c.c_int
_L = _c.c_long
_D = _c.c_double
_Z = _c.c_size_t

# name -> (restype, argtypes); one entry per function of include/nrgbd.h
SIGNATURES = {
    "nrgbd_version": (_c.c_char_p, []),
    "nrgbd_strerror": (_c.c_char_p, [_I]),
    "nrgbd_homography_terms": (_I, [_P, _P, _L, _L, _P, _L, _L, _P, _P, _I, _P]),
    "nrgbd_pose_inverse": (_I, [_P, _L, _P, _P, _I, _P]),
    "nrgbd_pack_nhwc": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P, _P]),
    "nrgbd_costvol_fwd": (_I, [_P, _P, _P, _P, _P, _P, _F, _F, _F, _I, _I, _P, _P,
                               _I, _I, _I, _I, _I, _I, _P]),
    "nrgbd_costvol_fwd_gen": (_I, [_P, _P, _P, _P, _P, _P, _F, _F, _F, _I, _I, _P, _P,
                                   _I, _I, _I, _I, _I, _I, _I, _P]),
    "nrgbd_costvol_bwd_workspace": (_I, [_I, _I, _I, _I, _I, _P]),
    "nrgbd_costvol_bwd": (_I, [_P, _P, _P, _P, _P, _P, _F, _F, _F, _I, _I, _P, _P, _P,
                               _I, _I, _I, _I, _I, _I, _P, _Z, _P]),
    "nrgbd_bn_cl_workgroups": (_I, [_L, _I]),
    "nrgbd_bn_cl_fwd": (_I, [_P, _P, _P, _P, _F, _F, _P, _P, _I, _P, _P, _P, _L, _I, _P]),
    "nrgbd_bn_cl_bwd": (_I, [_P, _P, _P, _I, _P, _P, _P, _P, _P, _L, _I, _P]),
    "nrgbd_warp_volume": (_I, [_P, _L, _L, _L, _L, _P, _L, _L, _L, _P, _P, _P, _P, _F, _F, _I,
                               _P, _P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "nrgbd_dpv_resample": (_I, [_P, _P, _P, _P, _F, _F, _F, _F, _F, _I, _F, _F, _P, _I, _I, _I, _P]),
    "nrgbd_dpv_resample_to": (_I, [_P, _P, _P, _P, _F, _F, _F, _F, _F, _I, _F, _F, _P, _I, _I, _I, _I, _P]),
    "nrgbd_logsoftmax_d": (_I, [_P, _P, _F, _P, _I, _L, _P]),
    "nrgbd_depth_regress": (_I, [_P, _P, _P, _P, _I, _L, _P]),
    "nrgbd_export_depth_u16": (_I, [_P, _P, _F, _F, _P, _P, _P, _P, _I, _L, _P]),
    "nrgbd_warp_depth_fwd": (_I, [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "nrgbd_warp_depth_bwd_workgroups": (_I, [_I, _I]),
    "nrgbd_warp_depth_bwd": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "nrgbd_conv3d_workgroups": (_I, [_I, _I, _I]),
    "nrgbd_conv3d_pack_weights": (_I, [_P, _P, _I, _P]),
    "nrgbd_conv3d_3x3x3_f32": (_I, [_P, _P, _I, _P, _P, _I, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "nrgbd_conv3d_3x3x3_cout1_f32": (_I, [_P, _P, _I, _P, _P, _I, _P, _P, _I, _I, _I, _I, _P]),
    "nrgbd_conv3d_cout1_dgrad_f32": (_I, [_P, _P, _P, _I, _I, _I, _P]),
    "nrgbd_conv3d_cout1_wgrad_workspace": (_I, [_I, _I, _I, _P]),
    "nrgbd_conv3d_cout1_wgrad_f32": (_I, [_P, _P, _P, _P, _Z, _I, _I, _I, _P]),
    "nrgbd_conv_wino_tiles": (_I, [_I, _I, _I, _I]),
    "nrgbd_conv_wino_rnet_f32": (_I, [_P, _P, _P, _I, _P, _I, _I, _I, _I, _I, _P]),
    "nrgbd_conv_wino_rnet_ex_f32": (_I, [_P, _P, _P, _I, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P]),
    "nrgbd_conv_wino_pack": (_I, [_P, _P, _I, _I, _I, _I, _P]),
    "nrgbd_bn_finalize_cm": (_I, [_P, _I, _I, _L, _P, _P, _F, _F, _P, _P, _P, _P, _P, _P]),
    "nrgbd_conv_wino_f32": (_I, [_P, _P, _I, _P, _P, _I, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P]),
    "nrgbd_conv_wino_dw_pack": (_I, [_P, _P, _I, _I, _I, _P]),
    "nrgbd_conv_wino_dw_f32": (_I, [_P, _P, _I, _P, _P, _I, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "nrgbd_conv_wino_dw_workgroups": (_I, [_I, _I, _I, _I]),
    "nrgbd_conv_wino_dw_unit_f32": (_I, [_P, _P, _F, _P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "nrgbd_conv_wino_dw4_pack": (_I, [_P, _P, _I, _I, _I, _P]),
    "nrgbd_conv_wino_dw4_workspace": (_I, [_I, _I, _I, _I, _P]),
    "nrgbd_conv_wino_dw4_f32": (_I, [_P, _P, _I, _F, _P, _P, _P, _P, _c.c_size_t, _I, _I, _I, _I, _I, _P]),
    "nrgbd_conv3d_wgrad_workgroups": (_I, []),
    "nrgbd_conv3d_wgrad_f32": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "nrgbd_bn3d_finalize": (_I, [_P, _I, _L, _P, _P, _F, _F, _P, _P, _P, _P, _P, _P]),
    "nrgbd_bn2d_partial_floats": (_I, [_I]),
    "nrgbd_bn2d_train_act": (_I, [_P, _P, _P, _F, _I, _P, _P, _P, _P, _I, _I, _L, _P]),
    "nrgbd_avgpool8": (_I, [_P, _P, _I, _I, _I, _P]),
    "nrgbd_conv2d_few_f32": (_I, [_P, _I, _P, _P, _I, _P, _I, _I, _I, _I, _I, _I, _I, _P]),
    "nrgbd_avgpool_cl": (_I, [_P, _P, _I, _I, _I, _I, _I, _P]),
    "nrgbd_scatter_channels": (_I, [_P, _L, _L, _L, _I, _I, _I, _P, _I, _I, _I, _L, _P]),
    "nrgbd_bias_act_nchw": (_I, [_P, _P, _F, _I, _I, _L, _P]),
    "nrgbd_conv2d_workgroups": (_I, [_I, _I, _I]),
    "nrgbd_conv2d_wgrad_workgroups": (_I, [_I, _I, _I, _I, _I]),
    "nrgbd_conv2d_wgrad_f32": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "nrgbd_conv2d_taps_f32": (_I, [_P, _P, _I, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P]),
    "nrgbd_space_to_depth2": (_I, [_P, _I, _P, _I, _I, _I, _I, _I, _P]),
    "nrgbd_conv_pack_weights": (_I, [_P, _P, _I, _I, _I, _P]),
    "nrgbd_conv2d_3x3_f32": (_I, [_P, _P, _I, _P, _P, _I, _P, _P, _P, _I, _P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "nrgbd_conv2d_rnet_f32": (_I, [_P, _P, _P, _I, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P]),
    "nrgbd_rnet_pack": (_I, [_P, _P, _I, _P, _I, _I, _L, _P]),
    "nrgbd_bn_finalize": (_I, [_P, _I, _I, _L, _P, _P, _F, _F, _P, _P, _P, _P, _P, _P]),
    "nrgbd_logsoftmax_rows": (_I, [_P, _P, _L, _I, _P]),
    "nrgbd_logsoftmax_d_bwd": (_I, [_P, _P, _F, _P, _I, _L, _P]),
    "nrgbd_logsoftmax_rows_bwd": (_I, [_P, _P, _P, _L, _I, _P]),
    "nrgbd_nll_workgroups": (_I, [_L]),
    "nrgbd_nll_fwd": (_I, [_P, _P, _L, _I, _L, _I, _P, _P, _P]),
    "nrgbd_nll_bwd": (_I, [_P, _L, _P, _P, _P, _I, _L, _I, _P]),
    "nrgbd_adam_step": (_I, [_P, _P, _P, _P, _P, _P, _I, _D, _D, _D, _D, _D, _I, _P]),
    "nrgbd_bias_lrelu_cl_workgroups": (_I, [_L, _I]),
    "nrgbd_bias_lrelu_cl_fwd": (_I, [_P, _P, _F, _P, _L, _I, _P]),
    "nrgbd_bias_lrelu_cl_bwd": (_I, [_P, _P, _F, _P, _P, _P, _L, _I, _P]),
    "nrgbd_upsample_bilinear_ac": (_I, [_P, _P, _I, _I, _I, _I, _I, _I, _I, _P]),
    "nrgbd_spp_concat": (_I, [_P, _I, _P, _I, _P, _P, _I, _I, _P, _P, _I, _I, _P, _P, _I, _I, _P, _P, _I, _I, _I, _P, _I, _I, _I, _P]),
    "nrgbd_nhwc_stats_workgroups": (_I, [_L]),
    "nrgbd_nhwc_stats": (_I, [_P, _L, _I, _P, _P]),
    "nrgbd_nhwc_act": (_I, [_P, _P, _I, _P, _P, _I, _P, _L, _I, _I, _P]),
}

_lib = None


class NrgbdError(RuntimeError):
    pass


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.isfile(LIB_PATH):
        raise NrgbdError(
            "libnrgbd_hip.so not found at %s — build it with `python -m neuralrgbd_amd.build` "
            "(hipcc --offload-arch=gfx950). There is no CPU / eager fallback for this path." % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError:
            raise NrgbdError("libnrgbd_hip.so does not export %s (stale build?)" % name)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(code, what):
    if code != 0:
        msg = load().nrgbd_strerror(int(code)).decode()
        raise NrgbdError("%s failed: %s (code %d)" % (what, msg, code))
