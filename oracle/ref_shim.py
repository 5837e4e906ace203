"""Import the UNMODIFIED reference (NVlabs/neuralrgbd at /root/reference) on CPU.

TEST INFRASTRUCTURE ONLY — used by oracle/gen_golden.py and by the `not gpu` tests that
pin the CPU oracle against the real reference.  /root/reference exists only in the build
container; on the GPU box `available()` is False and every user of this module skips.

Two shims are applied before the import:
  1. a stub `torchvision` module — code/mutils/misc.py:17 imports it and never uses it on
     this path;
  2. `.cuda()` becomes the identity and `torch.cuda.current_device()` returns 0 — the
     reference hard-codes `.cuda()` (warping/homography.py:306-311,440; models/KVNET.py:149).
Nothing under /root/reference is copied or modified.
"""
import contextlib
import io
import os
import sys
import types

REF_ROOT = os.environ.get("NRGBD_REFERENCE", "/root/reference")
REF_CODE = os.path.join(REF_ROOT, "code")


def available():
    return os.path.isfile(os.path.join(REF_CODE, "models", "KVNET.py"))


_loaded = None


def load():
    """Returns a namespace with the reference modules: KVNET, basic, homography, misc, test_step."""
    global _loaded
    if _loaded is not None:
        return _loaded
    if not available():
        raise RuntimeError("reference not present at %s" % REF_ROOT)
    import torch

    if "torchvision" not in sys.modules:
        sys.modules["torchvision"] = types.ModuleType("torchvision")
    if not torch.cuda.is_available():
        torch.Tensor.cuda = lambda self, *a, **k: self
        torch.nn.Module.cuda = lambda self, *a, **k: self
        torch.cuda.current_device = lambda: 0
    if REF_CODE not in sys.path:
        sys.path.insert(0, REF_CODE)
    with contextlib.redirect_stdout(io.StringIO()):
        import models.KVNET as m_kvnet
        import models.basic as m_basic
        import warping.homography as m_homo
        import warping.View as m_view
        import mutils.misc as m_misc
        import test_utils.test_KVNet as m_test
    _loaded = types.SimpleNamespace(KVNET=m_kvnet, basic=m_basic, homography=m_homo,
                                    View=m_view, misc=m_misc, test_step=m_test)
    return _loaded


@contextlib.contextmanager
def quiet():
    """The reference prints from constructors (KVNET.py:87-91, Refine.py:110-120)."""
    with contextlib.redirect_stdout(io.StringIO()):
        yield
