"""Builds libhgmm_hip.so (hand-written HIP kernels + C ABI) for gfx950, in-tree.

    python build.py [--force]

hipcc cross-compiles without a GPU.  The library is placed next to this file so that it
travels to the GPU box with the source snapshot (it is git-ignored, not gpurun-ignored).
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libhgmm_hip.so")
SOURCES = ["hgmm_api.hip", "flat_kernels.hip", "tree_kernels.hip", "tree_batch.hip", "kmeans_kernels.hip", "gmmreg_kernels.hip"]
HEADERS = ["hgmm_ctx.h", "wave_ops.h", "tree_device.h", os.path.join("..", "..", "include", "hgmm.h")]
ROCM = os.environ.get("ROCM_PATH", "/opt/rocm")
HIPCC = os.path.join(ROCM, "bin", "hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"]


def _newer(src, dst):
    return not os.path.exists(dst) or os.path.getmtime(src) > os.path.getmtime(dst)


def _compile(src):
    obj = os.path.join(CSRC, os.path.splitext(src)[0] + ".o")
    deps = [os.path.join(CSRC, src)] + [os.path.join(CSRC, h) for h in HEADERS]
    if not any(_newer(d, obj) for d in deps):
        return obj
    cmd = [HIPCC] + FLAGS + ["-c", os.path.join(CSRC, src), "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("hipcc failed for %s:\n%s\n%s" % (src, r.stdout, r.stderr))
    return obj


def build(force=False, verbose=True):
    if force:
        for s in SOURCES:
            o = os.path.join(CSRC, os.path.splitext(s)[0] + ".o")
            if os.path.exists(o):
                os.remove(o)
    with ThreadPoolExecutor(max_workers=len(SOURCES)) as ex:
        objs = list(ex.map(_compile, SOURCES))
    if force or any(_newer(o, LIB) for o in objs):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs + \
              ["-L" + os.path.join(ROCM, "lib"), "-lrccl", "-Wl,-rpath," + os.path.join(ROCM, "lib")]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n%s\n%s" % (r.stdout, r.stderr))
        if verbose:
            print("built", LIB)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
