"""Build libdeephar_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU).

    python -m deephar_amd.csrc.build [--force]

The .so is git-ignored but travels with gpurun snapshots.  Objects are rebuilt only when a source or
header is newer (hipcc takes ~15-25 s per kernel file).
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
INCLUDE = os.path.join(ROOT, "include")
SOURCES = ["conv_igemm.hip", "gemm1x1.hip", "gemm1x1s.hip", "conv_splitk.hip", "conv_halo.hip", "conv_stem.hip", "spatial.hip", "decoder.hip",
           "capi.hip", "plan.hip"]
HEADERS = [os.path.join(HERE, "dh_kernels.h"), os.path.join(HERE, "conv_common.h"), os.path.join(HERE, "dw_lds.h"),
           os.path.join(INCLUDE, "deephar_hip.h")]
LIB = os.path.join(HERE, "libdeephar_hip.so")
OBJDIR = os.path.join(HERE, "build")
ARCH = "gfx950"
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-I" + INCLUDE, "-I" + HERE,
         "-Wno-unused-result"]


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _compile(src):
    obj = os.path.join(OBJDIR, src.replace(".hip", ".o"))
    path = os.path.join(HERE, src)
    if _stale(obj, [path] + HEADERS):
        cmd = [HIPCC] + FLAGS + ["-c", path, "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed for %s:\n%s\n%s" % (src, r.stdout, r.stderr))
    return obj


def build(force=False, verbose=True):
    os.makedirs(OBJDIR, exist_ok=True)
    if force:
        for f in os.listdir(OBJDIR):
            os.remove(os.path.join(OBJDIR, f))
        if os.path.exists(LIB):
            os.remove(LIB)
    with ThreadPoolExecutor(max_workers=len(SOURCES)) as ex:
        objs = list(ex.map(_compile, SOURCES))
    if _stale(LIB, objs):
        cmd = [HIPCC, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", LIB] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n%s\n%s" % (r.stdout, r.stderr))
        if verbose:
            print("built", LIB)
    elif verbose:
        print("up to date:", LIB)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
