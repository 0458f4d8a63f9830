"""Build libalignn_hip.so (gfx950) in-tree with hipcc.  ``python -m alignn_amd.build``.

Every source is compiled to its own object (in parallel, only when it or a header changed) and the objects are linked
into ONE shared library - the same command line per file as a single hipcc invocation, but an edit of one kernel file
rebuilds in seconds."""

from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "build")
LIB = os.path.join(HERE, "libalignn_hip.so")
SOURCES = ["norm.hip", "conv.hip", "gemm_f32.hip", "gemm_x6.hip", "embed.hip", "dual.hip", "knn.hip", "composite.hip", "model.hip", "stage.hip", "radius.hip", "angle.hip", "ff.hip", "convln.hip", "gemm_dw.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC"]
# Per-file additions.  norm.hip / dual.hip (the LayerNorm kernels and their dual-number twins): no SLP vectorisation, i.e. no
# packed-fp32 instructions (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32 with op_sel).  With them hipcc 7.2's code for
# ln_silu_bwd_kernel intermittently returned one float4 component of lanes 48-63 wrong when waves of ANOTHER kernel (an MFMA
# projection on a second stream) shared the compute unit - run-to-run different forces and gradients of ALIGNNAtomWise on lanes;
# compiled without them (or at -O1) every run is bit-identical (round 5: DESIGN.md section 4e, profiles/r05_ln_concurrency.txt).
# The streaming kernels of these two files are memory-bound: no measurable cost (headline 14.89 vs 14.84-14.89 ms, force training
# 37.30 vs 37.24-37.28 ms, same box).  The whole library without SLP costs the headline 0.55 ms (projection / gate epilogues).
EXTRA_FLAGS = {"norm.hip": ["-fno-slp-vectorize"], "dual.hip": ["-fno-slp-vectorize"], "convln.hip": ["-fno-slp-vectorize"]}


def _headers():
    return [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")] + [
        os.path.join(os.path.dirname(HERE), "include", "alignn_hip.h")]


def _stale(target, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False) -> str:
    """Compile every HIP source for gfx950 into one shared library; returns its path."""
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    hdrs = _headers()
    srcs = [s for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    if not force and not _stale(LIB, [os.path.join(CSRC, s) for s in srcs] + hdrs + [os.path.abspath(__file__)]):
        return LIB  # (e.g. on the GPU box: the prebuilt library travelled with the snapshot, the objects did not)
    os.makedirs(OBJ, exist_ok=True)
    objs = [os.path.join(OBJ, s.replace(".hip", ".o")) for s in srcs]

    def compile_one(pair):
        src, obj = pair
        if force or _stale(obj, [os.path.join(CSRC, src), os.path.abspath(__file__)] + hdrs):
            cmd = [hipcc] + FLAGS + EXTRA_FLAGS.get(src, []) + ["-c", os.path.join(CSRC, src), "-o", obj]
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.run(cmd, check=True)
            return True
        return False

    with ThreadPoolExecutor(max_workers=min(8, len(srcs))) as pool:
        rebuilt = list(pool.map(compile_one, zip(srcs, objs)))
    if force or any(rebuilt) or _stale(LIB, objs):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
