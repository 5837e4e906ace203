"""neuralrgbd_amd — MI355X-native (gfx950) plane-sweep depth path of NVlabs/neuralrgbd.

Drop-in surface (reference file it replaces):
    neuralrgbd_amd.KVNET            code/models/KVNET.py::KVNET
    neuralrgbd_amd.homography       code/warping/homography.py (the operators on the hot path)
    neuralrgbd_amd.test_step.test   code/test_utils/test_KVNet.py::test
    neuralrgbd_amd.camera           the cam_intrinsics dict of code/mdataloader/scanNet.py:204-272
The compute between the convolutions is hand-written HIP behind the C-ABI of include/nrgbd.h
(libnrgbd_hip.so, built by `python -m neuralrgbd_amd.build`).
"""
import os as _os

# ROCm's hipGraph "packet capture" (pre-recorded AQL packets for kernel nodes, on by default in CLR) is switched off for this process
# unless the caller decided otherwise: with it on, replays of a captured TRAINING iteration came out wrong — deterministically, by
# 0.1-0.3 % of the loss — whenever eager launches of the same kernels had run between the replays (an eager twin in a test, an evaluation
# pass between optimizer steps); serialised launches or this switch remove it, the measured cost is nil (train 29.19 vs 29.24 ms per
# window, config B 30.62 vs 30.64 frames/s, config S 204.4 vs 203.8: DESIGN.md section 7).  The runtime reads the variable when it
# initialises, i.e. at the process's first HIP call: importing this package before touching the GPU is enough.
_os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")

from . import camera, synth  # host-only helpers (numpy / torch CPU)


def __getattr__(name):
    # the GPU-facing modules load libnrgbd_hip.so; import them lazily so that host-only tooling
    # (camera, synth, build) works before the library is built
    if name in ("ops", "homography", "kvnet", "nets", "misc", "test_step", "_lib"):
        import importlib
        return importlib.import_module("." + name, __name__)
    if name == "KVNET":
        from .kvnet import KVNET
        return KVNET
    raise AttributeError(name)
