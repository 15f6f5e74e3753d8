#!/usr/bin/env python3
"""Builds A/B variants of the library into build/ab/lib_<TAG>.so:  tools/ab_build.py [--only file.hip] TAG=-DX=1,-DY ...
--only: the flags go to that one source (its object is rebuilt), every other object is the default build's -- a minute instead of three
when the switch lives in one kernel file."""
import os, subprocess, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from coper_amd import build
os.makedirs(os.path.join(build.HERE, "..", "build", "ab"), exist_ok=True)
args = sys.argv[1:]
only = None
if args and args[0] == "--only":
    only, args = args[1], args[2:]
for spec in args:
    tag, _, flags = spec.partition("=")
    flags = [f for f in flags.split(",") if f]
    if any(f.startswith("-DCOPER_DBG_") for f in flags) and "-DCOPER_DIAG" not in flags:
        flags.append("-DCOPER_DIAG")            # (the ablation switches exist in diagnostic builds only: csrc/coper_internal.h)
    out = os.path.join(build.HERE, "..", "build", "ab", "lib_%s.so" % tag)
    if only is None:
        build.build_library(force=True, extra_flags=flags, out=out)
    else:
        build.build_library()                                   # the default objects, up to date
        objdir = os.path.join(build.HERE, "..", "build", "obj", "default")
        ab_obj = os.path.join(build.HERE, "..", "build", "ab", "%s.%s.o" % (tag, only))
        base = [build._hipcc(), "-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-fvisibility=hidden", "-Wall",
                "-Wno-unused-function", "-DCOPER_BUILD", *flags]
        subprocess.check_call(base + build.SOURCE_FLAGS.get(only, []) + ["-c", os.path.join(build.CSRC, only), "-o", ab_obj])
        objs = [ab_obj if s == only else os.path.join(objdir, s + ".o") for s in build.SOURCES]
        subprocess.check_call([build._hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out, *objs])
    print("built", out)
