"""Build the gfx950 HIP library in-tree: powerserve_amd/lib/libps_hip.so (hipcc cross-compiles without a GPU)."""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
OBJDIR = os.path.join(HERE, "build")
SOURCES = ["api.hip", "model.hip", "k_quant.hip", "k_gemv.hip", "k_gemv4.hip", "k_qkvattn.hip", "k_gemvb.hip", "k_gemvk.hip", "k_gemm4k.hip", "k_gemv6.hip", "k_ops.hip", "k_attn.hip", "perf16.hip"]
# -ffp-contract=off: the parity contract needs every fp32 op to round where the reference's C source rounds;
# fused multiply-adds are written explicitly (__fmaf_rn) where the reference uses FMA intrinsics.
# -fno-slp-vectorize (NOSLP files only): hipcc's SLP vectoriser packs adjacent scalar fp32 operations into v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32.  On
# gfx950 a scalar v_fma_f32 already runs at the packed rate (2 cycles per wave64, MI355X_MICROARCH.md) and the packed forms cost more in dependent chains
# (the mat-vec's chain wave) and beside matrix instructions: the same source without them, same bits, 8B decode 543.6 -> 552.3 tok/s
# (profiles/r04_prefill_ab.txt).  Only the files of the single-token path take the flag.  (With it on every file one 8B Q5_K_M bench
# run did not come back within 15 minutes; the likeliest culprit is that run's CPU baseline leg -- the reference's spin-barrier thread pool on a busy
# host, now a child process under a time limit in bench.py -- but the round had no GPU time left to tell the two apart, so every other file keeps the
# exact compile configuration that has passed the full GPU suite and all bench configurations.)
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Wno-unused-result",
         "-I" + os.path.join(HERE, "..", "include")]
NOSLP = {"k_gemv4.hip", "k_qkvattn.hip", "k_gemvb.hip", "k_attn.hip", "k_gemm4k.hip"}  # k_gemm4k.hip (round 5): gate/up chunk launch of a 512-column sequence 384.2 -> 369-371 us, same bits (profiles/r05_g4k2_variants.txt, the "v2 0" lines vs profiles/r05_g4k_item_order.txt)  # (k_gemvb.hip: the Q4_0 / Q8_0 mat-vec, Llama-3.2-1B 1494 -> 1525 tok/s)


def _cmd_changed(obj: str, cmd: list) -> bool:
    """the exact compile command is kept next to the object: an edited flag list rebuilds it (mtimes alone let stale objects survive)"""
    try:
        return open(obj + ".cmd").read() != " ".join(cmd)
    except OSError:
        return True


def _newer(src: str, obj: str) -> bool:
    if not os.path.exists(obj):
        return True
    t = os.path.getmtime(obj)
    deps = [src] + [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    deps.append(os.path.join(HERE, "..", "include", "ps_hip.h"))
    return any(os.path.getmtime(d) > t for d in deps)


# The files that hold one of the reference's fp-contraction sites (ps_dev.h: ps_rope_pair / ps_rope_one / ps_dot_left, and the Q5_K refusal): only they are
# compiled a second time for lib/libps_hip_contract.so (-DPS_CONTRACT: the reference's stock -ffp-contract=fast build, include/ps_hip.h ps_hip_build_contract);
# every other object is shared with the default library.
CONTRACT_SOURCES = {"api.hip", "k_ops.hip", "k_attn.hip", "k_gemm4k.hip", "k_gemv4.hip", "k_qkvattn.hip", "k_gemvb.hip", "k_gemvk.hip"}


# Files with in-kernel timeline marks (ps_dev.h PS_TIMELINE): compiled a third time, on demand, for lib/libps_hip_timeline.so -- the library the timeline tools load.
TIMELINE_SOURCES = {"api.hip", "k_gemv4.hip", "k_qkvattn.hip", "k_attn.hip", "k_gemm4k.hip"}


def build(force: bool = False, verbose: bool = True, contract: bool = True, timeline: bool = False) -> str:
    os.makedirs(LIBDIR, exist_ok=True)
    os.makedirs(OBJDIR, exist_ok=True)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objs, cobjs, tobjs, jobs = [], [], [], []
    for s in SOURCES:
        src, obj = os.path.join(CSRC, s), os.path.join(OBJDIR, s.replace(".hip", ".o"))
        objs.append(obj)
        cmd = [hipcc, *FLAGS, *(["-fno-slp-vectorize"] if s in NOSLP else []), "-c", src, "-o", obj]
        if force or _newer(src, obj) or _cmd_changed(obj, cmd):
            jobs.append(cmd)
        if contract and s in CONTRACT_SOURCES:
            cobj = os.path.join(OBJDIR, s.replace(".hip", ".contract.o"))
            ccmd = [hipcc, *FLAGS, *(["-fno-slp-vectorize"] if s in NOSLP else []), "-DPS_CONTRACT=1", "-c", src, "-o", cobj]
            if force or _newer(src, cobj) or _cmd_changed(cobj, ccmd):
                jobs.append(ccmd)
            cobjs.append(cobj)
        else:
            cobjs.append(obj)
        if timeline and s in TIMELINE_SOURCES:
            tobj = os.path.join(OBJDIR, s.replace(".hip", ".timeline.o"))
            tcmd = [hipcc, *FLAGS, *(["-fno-slp-vectorize"] if s in NOSLP else []), "-DPS_TIMELINE=1", "-c", src, "-o", tobj]
            if force or _newer(src, tobj) or _cmd_changed(tobj, tcmd):
                jobs.append(tcmd)
            tobjs.append(tobj)
        else:
            tobjs.append(obj)

    def run(cmd):
        if verbose:
            print("[build]", os.path.basename(cmd[-1]), flush=True)
        subprocess.run(cmd, check=True)
        open(cmd[-1] + ".cmd", "w").write(" ".join(cmd))

    with ThreadPoolExecutor(max_workers=8) as ex:
        list(ex.map(run, jobs))
    so = os.path.join(LIBDIR, "libps_hip.so")
    if jobs or not os.path.exists(so):
        subprocess.run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", so, *objs], check=True)
    cso = os.path.join(LIBDIR, "libps_hip_contract.so")
    if contract and (jobs or not os.path.exists(cso)):
        subprocess.run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", cso, *cobjs], check=True)
    tso = os.path.join(LIBDIR, "libps_hip_timeline.so")
    if not timeline and jobs and os.path.exists(tso):
        os.remove(tso)  # (a kernel was rebuilt without its timeline twin: the timeline tools prefer that library when it exists and would time an OLD kernel)
    if timeline and (jobs or not os.path.exists(tso)):
        subprocess.run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", tso, *tobjs], check=True)
    build_host(force or bool(jobs), verbose)
    return so


HOST_SOURCES = ["json_gguf.cpp", "graph.cpp", "hip_backend.cpp", "model.cpp", "speculative.cpp", "sampler.cpp"]


def build_host(force: bool = False, verbose: bool = True) -> str:
    """C++20 host facade (Graph / Executor / HIPBackend / Model mirrors) -> lib/libps_host.so on top of the C-ABI.
    POWERSERVE_EXCEPTION_ABORT: failures throw (reference option, src/core/exception.hpp:275-278) so that the
    ctypes driver can report them instead of aborting the interpreter."""
    hdir = os.path.join(CSRC, "host")
    so = os.path.join(LIBDIR, "libps_host.so")
    srcs = [os.path.join(hdir, f) for f in HOST_SOURCES]
    deps = srcs + [os.path.join(hdir, f) for f in os.listdir(hdir) if f.endswith(".hpp")] + [os.path.join(HERE, "..", "include", "ps_hip.h")]
    if not force and os.path.exists(so) and all(os.path.getmtime(d) <= os.path.getmtime(so) for d in deps):
        return so
    if verbose:
        print("[build] host facade", flush=True)
    subprocess.run(["g++", "-std=c++20", "-O2", "-fPIC", "-shared", "-DPOWERSERVE_EXCEPTION_ABORT", "-Wall", "-Wno-unused-function",
                    "-o", so, *srcs, "-L" + LIBDIR, "-lps_hip", "-Wl,-rpath,$ORIGIN"], check=True)
    return so


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, timeline="--timeline" in sys.argv))
