"""Builds libcoper_hip.so (gfx950) in-tree with hipcc.  No torch headers: the library is a plain
C-ABI shared object (include/coper_hip.h)."""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB_PATH = os.path.join(HERE, "libcoper_hip.so")
SOURCES = ["coper_abi.hip", "kernels_prepare.hip", "kernels_encode.hip", "kernels_score.hip", "kernels_score_bf16.hip",
           "kernels_encode_bf16.hip"]


def _hipcc():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (set HIPCC)")


def needs_build():
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(HERE, "..", "include", "coper_hip.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build_library(force=False, verbose=False, extra_flags=()):
    if not force and not needs_build():
        return LIB_PATH
    cmd = [_hipcc(), "-O3", "-std=c++17", "--offload-arch=gfx950", "-shared", "-fPIC",
           "-fvisibility=hidden", "-Wall", "-Wno-unused-function",
           # accumulate MFMAs in place in VGPRs: without it hipcc parks accumulators in AGPRs and shuffles
           # them through a working range with v_accvgpr_mov around every chain (dense kernels: -40 % VGPRs)
           "-mllvm", "-amdgpu-mfma-vgpr-form",
           "-DCOPER_BUILD", *extra_flags, "-o", LIB_PATH + ".tmp"]
    cmd += [os.path.join(CSRC, s) for s in SOURCES]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    os.replace(LIB_PATH + ".tmp", LIB_PATH)
    return LIB_PATH


if __name__ == "__main__":
    print(build_library(force="--force" in sys.argv, verbose=True))
