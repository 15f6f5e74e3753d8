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
SOURCES = ["coper_abi.hip", "kernels_prepare.hip", "kernels_encode.hip", "kernels_score.hip", "kernels_score_bf16.hip", "kernels_score3_bf16.hip", "kernels_tail_bf16.hip",
           "kernels_encode_bf16.hip", "kernels_dense_fused_bf16.hip", "kernels_topk_bf16.hip", "coper_train.hip", "train_gemm_bf16.hip", "train_gemm_w128_bf16.hip", "sampler.hip"]


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


# per-source extra flags
#  -amdgpu-mfma-vgpr-form: accumulate MFMAs in place in VGPRs: without it hipcc parks accumulators in AGPRs and
#  shuffles them through a working range with v_accvgpr_mov around every chain (dense kernels: -40 % VGPRs)
VGPR_FORM = ["-mllvm", "-amdgpu-mfma-vgpr-form"]
SOURCE_FLAGS = {name: VGPR_FORM for name in SOURCES}
# the fused conv + dense kernel wants its 128 accumulators in AGPRs: the conv's taps and the weight prefetch fill the VGPRs
SOURCE_FLAGS["kernels_dense_fused_bf16.hip"] = []
# the pipelined score kernel keeps its accumulators in AGPRs and everything else in the 256 VGPRs (one wave per SIMD)
SOURCE_FLAGS["kernels_score3_bf16.hip"] = []
SOURCE_FLAGS["train_gemm_w128_bf16.hip"] = []   # 256 accumulator registers per lane: they have to be AGPRs


def _compile_one(args):
    cmd, verbose = args
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)


def build_library(force=False, verbose=False, extra_flags=(), out=None, jobs=4):
    """Compiles each source to an object under build/obj/<flags-key>/ (only the stale ones), then links."""
    out = out or LIB_PATH
    if not force and out == LIB_PATH and not extra_flags and not needs_build():
        return LIB_PATH
    import hashlib
    from concurrent.futures import ThreadPoolExecutor
    key = hashlib.sha1(" ".join(extra_flags).encode()).hexdigest()[:10] if extra_flags else "default"
    objdir = os.path.join(HERE, "..", "build", "obj", key)
    os.makedirs(objdir, exist_ok=True)
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".h", ".inc"))]
    headers.append(os.path.join(HERE, "..", "include", "coper_hip.h"))
    hdr_t = max(os.path.getmtime(p) for p in headers + [os.path.abspath(__file__)])
    base = [_hipcc(), "-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-fvisibility=hidden", "-Wall",
            "-Wno-unused-function", "-DCOPER_BUILD", *extra_flags]
    todo, objs = [], []
    for src in SOURCES:
        sp = os.path.join(CSRC, src)
        op = os.path.join(objdir, src + ".o")
        objs.append(op)
        if os.path.exists(op) and os.path.getmtime(op) > max(os.path.getmtime(sp), hdr_t):
            continue
        todo.append((base + SOURCE_FLAGS.get(src, []) + ["-c", sp, "-o", op], verbose))
    with ThreadPoolExecutor(max_workers=jobs) as ex:
        list(ex.map(_compile_one, todo))
    link = [_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out + ".tmp", *objs]
    if verbose:
        print(" ".join(link), file=sys.stderr)
    subprocess.check_call(link)
    os.replace(out + ".tmp", out)
    return out


if __name__ == "__main__":
    print(build_library(force="--force" in sys.argv, verbose=True))
