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


OBJ = os.path.join(CSRC, "_obj")


def _headers():
    return glob.glob(os.path.join(CSRC, "*.hpp")) + glob.glob(os.path.join(INCLUDE, "*.h"))


def stale():
    if not os.path.isfile(LIB):
        return True
    deps = sources() + _headers()
    return any(os.path.getmtime(d) > os.path.getmtime(LIB) for d in deps)


def _compile(src, verbose):
    """One translation unit -> object file (only when the source or any header is newer)."""
    obj = os.path.join(OBJ, os.path.basename(src)[:-4] + ".o")
    newest = max(os.path.getmtime(d) for d in [src] + _headers())
    if os.path.isfile(obj) and os.path.getmtime(obj) >= newest:
        return obj
    cmd = [HIPCC] + [f for f in FLAGS if f != "-shared"] + ["-c", "-I", INCLUDE, src, "-o", obj]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return obj


def build_dev(verbose=False):
    """Developer variant libnrgbd_hip_dev.so (-DNRGBD_DEV: ablation bits and tile-order switches honoured, read from the
    environment).  Used only by tools/ for kernel analysis; never loaded by the package."""
    lib = os.path.join(CSRC, "libnrgbd_hip_dev.so")
    extra = os.environ.get("NRGBD_DEV_DEFINES", "").split()      # e.g. NRGBD_DEV_DEFINES="-DNRGBD_DW_NT=1" for a one-off experiment
    cmd = [HIPCC] + FLAGS + ["-DNRGBD_DEV"] + extra + ["-I", INCLUDE] + sources() + ["-o", lib]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return lib


def build_variant(name, defines, only=("wino_dw.hip",), verbose=False):
    """Experimental A/B library libnrgbd_exp_<name>.so: the product objects with `only` recompiled under extra -D defines.
    Git-ignored; loaded by tools/ through _lib.LIB_PATH, never by the package."""
    build()
    lib = os.path.join(CSRC, "libnrgbd_exp_%s.so" % name)
    objs = []
    for src in sources():
        base = os.path.basename(src)
        if base in only:
            obj = os.path.join(OBJ, "%s.%s.o" % (base[:-4], name))
            cmd = [HIPCC] + [f for f in FLAGS if f != "-shared"] + list(defines) + ["-c", "-I", INCLUDE, src, "-o", obj]
            if verbose:
                print(" ".join(cmd))
            subprocess.check_call(cmd)
        else:
            obj = os.path.join(OBJ, base[:-4] + ".o")
        objs.append(obj)
    subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", lib])
    return lib


def build(force=False, verbose=False):
    """Per-file objects (compiled in parallel, re-used when unchanged) linked into libnrgbd_hip.so."""
    if not force and not stale():
        return LIB
    from concurrent.futures import ThreadPoolExecutor
    os.makedirs(OBJ, exist_ok=True)
    if force:
        for f in glob.glob(os.path.join(OBJ, "*.o")):
            os.remove(f)
    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as ex:
        objs = list(ex.map(lambda s_: _compile(s_, verbose), sources()))
    cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", LIB + ".tmp"]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    os.replace(LIB + ".tmp", LIB)
    return LIB


if __name__ == "__main__":
    if "--dev" in sys.argv:
        print(build_dev(verbose=True))
    else:
        print(build(force="--force" in sys.argv, verbose=True))
