"""Builds text_similarity_amd/libtsim.so (the C-ABI HIP library) in-tree for gfx950.

    python -m text_similarity_amd.build [--force]

hipcc cross-compiles without a GPU; the .so is git-ignored but travels to the GPU box with the tree.
"""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
# TSIM_BUILD_TAG=<tag>: a variant build beside the product (objects in csrc/_obj_<tag>, library libtsim_<tag>.so; load it with
# TSIM_LIB=<path>).  Used for the diagnostic builds with in-kernel stamps and for A/B variants shipped to the GPU box together.
TAG = os.environ.get("TSIM_BUILD_TAG", "")
OBJ = os.path.join(CSRC, "_obj" + ("_" + TAG if TAG else ""))
LIB = os.path.join(HERE, "libtsim" + ("_" + TAG if TAG else "") + ".so")
SOURCES = ["common.hip", "search.hip", "k1_kl16.hip", "k1_d384.hip", "k1_kl32.hip", "k1_collect.hip", "gemm_pp.hip", "encoder.hip",
           "wordpiece.cpp"]   # (host-only C++: the ASCII WordPiece tokenizer)
HOT_KERNELS = ("cos_topk_partial", "cos_topk_finalize", "gemm_bf16", "gemm_xres", "ln_rows_gemm", "ln_tail_gemm", "gemm_pp", "attention_kernel")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-gpu-rdc", "-Wall", "-Wno-unused-function",
         "-I" + os.path.join(os.path.dirname(HERE), "include")]


def _newer(a, bs):
    return os.path.exists(a) and all(os.path.getmtime(a) >= os.path.getmtime(b) for b in bs)


# Timing-only switches that make kernels return WRONG results (they cut work out to see what it costs).  They may only go
# into a variant build (TSIM_BUILD_TAG set -> libtsim_<tag>.so), never into the product library.
DIAGNOSTIC_DEFINES = ("TSIM_LN_DIAG", "TSIM_X2_DIAG", "TSIM_FF_DIAG", "TSIM_K1_NOSEL", "TSIM_PP_STAMPS", "TSIM_K1_DIAG",
                      "TSIM_ATT_DIAG", "TSIM_L_DIAG")


def build(force: bool = False, verbose: bool = True, extra_flags=(), only=None) -> str:
    """``only``: compile just these sources (the others must have objects already) — quick iteration on one kernel."""
    bad = [f for f in extra_flags if f.startswith("-D") and ("DIAG" in f or any(d in f for d in DIAGNOSTIC_DEFINES))]
    if bad and not TAG:
        raise RuntimeError(f"{bad}: diagnostic defines produce wrong results; they are refused for the product library — "
                           "set TSIM_BUILD_TAG=<tag> to build libtsim_<tag>.so beside it")
    os.makedirs(OBJ, exist_ok=True)
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    hdrs.append(os.path.join(os.path.dirname(HERE), "include", "tsim.h"))
    srcs = [s for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    jobs = []
    for s in srcs:
        src = os.path.join(CSRC, s)
        obj = os.path.join(OBJ, os.path.splitext(s)[0] + ".o")
        if only is not None and s not in only:
            if not os.path.exists(obj):
                raise RuntimeError(f"--only: {obj} does not exist yet")
            continue
        if force or not _newer(obj, [src] + hdrs):
            jobs.append((src, obj))

    def cc(job):
        src, obj = job
        cmd = ["hipcc", *FLAGS, *extra_flags, "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, stderr=subprocess.PIPE, text=True)
        if r.returncode != 0 or "warning:" in r.stderr or "error:" in r.stderr:
            # remarks carry source-context lines; show everything except the remark blocks
            keep, skip = [], 0
            for ln in r.stderr.splitlines():
                if "remark:" in ln:
                    skip = 2
                elif skip:
                    skip -= 1
                else:
                    keep.append(ln)
            print("\n".join(keep), file=sys.stderr, flush=True)
        if r.returncode != 0:
            raise subprocess.CalledProcessError(r.returncode, cmd)
        # a hot kernel that spills to scratch is a silent 10x slowdown (it happened: nested lambdas that stopped being
        # inlined put the resident MFMA fragments in memory) -> fail the build instead
        name = None
        for ln in r.stderr.splitlines():
            if "Function Name:" in ln:
                name = ln.split("Function Name:")[1].split("[")[0].strip()
            elif "ScratchSize [bytes/lane]:" in ln and name:
                n = int(ln.split("ScratchSize [bytes/lane]:")[1].split("[")[0])
                # (a tagged variant build may opt out: TSIM_ALLOW_SCRATCH=1 — in-kernel stamps cost a few registers)
                if n > 0 and any(k in name for k in HOT_KERNELS) and not (TAG and os.environ.get("TSIM_ALLOW_SCRATCH")):
                    try:
                        os.remove(obj)
                    except OSError:
                        pass
                    raise RuntimeError(f"{os.path.basename(src)}: hot kernel {name} uses {n} B/lane of scratch")

    with ThreadPoolExecutor(max_workers=4) as ex:
        list(ex.map(cc, jobs))
    objs = [os.path.join(OBJ, os.path.splitext(s)[0] + ".o") for s in srcs]
    if force or jobs or not _newer(LIB, objs):
        cmd = ["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-pthread", "-o", LIB, *objs]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
        # every kernel's host stub must be defined (hipcc has silently dropped one: an undefined symbol only shows at
        # dlopen time, i.e. on the GPU box) -> load the library once in a child process right after linking
        chk = subprocess.run([sys.executable, "-c", f"import ctypes; ctypes.CDLL({LIB!r})"], stderr=subprocess.PIPE, text=True)
        if chk.returncode != 0:
            raise RuntimeError(f"{LIB} does not load: {chk.stderr.strip().splitlines()[-1] if chk.stderr.strip() else chk.returncode}")
    return LIB


if __name__ == "__main__":
    # --stamps: DIAGNOSTIC build of the ping-pong GEMM with in-kernel cycle stamps (tools/pp_stamps.py); rebuild without it after
    # -DNAME arguments are passed through to hipcc (A/B builds of compile-time variants; they force a rebuild)
    defs = [a for a in sys.argv[1:] if a.startswith("-D")]
    only = [f for a in sys.argv[1:] if a.startswith("--only=") for f in a.split("=", 1)[1].split(",")]
    build(force="--force" in sys.argv or "--stamps" in sys.argv or bool(defs),
          extra_flags=(["-DTSIM_PP_STAMPS"] if "--stamps" in sys.argv else []) + defs, only=only or None)
    print(LIB)
