"""neuralrgbd_amd — MI355X-native (gfx950) plane-sweep depth path of NVlabs/neuralrgbd.

Drop-in surface (reference file it replaces):
    neuralrgbd_amd.KVNET            code/models/KVNET.py::KVNET
    neuralrgbd_amd.homography       code/warping/homography.py (the operators on the hot path)
    neuralrgbd_amd.test_step.test   code/test_utils/test_KVNet.py::test
    neuralrgbd_amd.camera           the cam_intrinsics dict of code/mdataloader/scanNet.py:204-272
The compute between the convolutions is hand-written HIP behind the C-ABI of include/nrgbd.h
(libnrgbd_hip.so, built by `python -m neuralrgbd_amd.build`).
"""
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
