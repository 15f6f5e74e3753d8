#!/usr/bin/env python3
"""Builds A/B variants of the library into build/ab/lib_<TAG>.so:  tools/ab_build.py TAG=-DX=1,-DY ..."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from coper_amd import build
os.makedirs(os.path.join(build.HERE, "..", "build", "ab"), exist_ok=True)
for spec in sys.argv[1:]:
    tag, _, flags = spec.partition("=")
    out = os.path.join(build.HERE, "..", "build", "ab", "lib_%s.so" % tag)
    build.build_library(force=True, extra_flags=[f for f in flags.split(",") if f], out=out)
    print("built", out)
