#!/usr/bin/env python3
"""A/B build of the HIP library for one-constant experiments: recompile the named sources with extra -D flags and link them with the
production objects into powerserve_amd/lib/libps_hip_<name>.so; run with PS_HIP_LIB=<that path> (powerserve_amd/hip.py).
usage: ab_build.py NAME file.hip[,file.hip...] -DX=1 [-DY=2 ...]"""
import os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from powerserve_amd import build as B

name, files, defs = sys.argv[1], sys.argv[2].split(","), sys.argv[3:]
B.build(verbose=False)
objs = []
for s in B.SOURCES:
    obj = os.path.join(B.OBJDIR, s.replace(".hip", ".o"))
    if s in files:
        obj = os.path.join(B.OBJDIR, s.replace(".hip", f"_{name}.o"))
        subprocess.run([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc"), *B.FLAGS, *(["-fno-slp-vectorize"] if s in B.NOSLP else []), *defs, "-c", os.path.join(B.CSRC, s), "-o", obj], check=True)
    objs.append(obj)
so = os.path.join(B.LIBDIR, f"libps_hip_{name}.so")
subprocess.run([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc"), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", so, *objs], check=True)
print(so)
