"""Build libnrgbd_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU).

    python -m neuralrgbd_amd.build [--force]

The shared object is written next to the sources (neuralrgbd_amd/csrc/) so that it travels
with the tree; there is no JIT cache and no pip install.
"""
import glob
import os
import subprocess
import sys

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
INCLUDE = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include")
LIB = os.path.join(CSRC, "libnrgbd_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared",
         "-Wno-comment"]


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def stale():
    if not os.path.isfile(LIB):
        return True
    deps = sources() + glob.glob(os.path.join(CSRC, "*.hpp")) + glob.glob(os.path.join(INCLUDE, "*.h"))
    return any(os.path.getmtime(d) > os.path.getmtime(LIB) for d in deps)


def build(force=False, verbose=False):
    if not force and not stale():
        return LIB
    cmd = [HIPCC] + FLAGS + ["-I", INCLUDE] + sources() + ["-o", LIB + ".tmp"]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    os.replace(LIB + ".tmp", LIB)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
