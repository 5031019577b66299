"""Build libgscan_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU).

    python -m multimodal_seq2seq_gscan_amd.build [--force] [--verbose]

The shared object lands next to this file so that it travels with the source tree; it is
git-ignored.  Objects are rebuilt only when a source or header is newer."""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
INCLUDE = os.path.join(os.path.dirname(HERE), "include")
OBJ = os.path.join(CSRC, "build")
LIB = os.path.join(HERE, "libgscan_hip.so")
SOURCES = ["gemm.hip", "gemm_mt.hip", "gemm_ws.hip", "conv.hip", "elementwise.hip", "loss.hip", "lstm_encoder.hip", "decoder.hip", "decoder_any.hip", "attention_grad.hip", "step.hip",
           "probe.hip", "comm.hip", "capi.hip"]
# decoder.hip is compiled as four translation units, a quarter of the hidden sizes each (-DGSCAN_DEC_PART=k, csrc/step.h):
# on one core the file took 80 s of the build's 95
PARTS = {"decoder.hip": 4}


def units():
    """(source file, object name, extra flags) of every translation unit of the library"""
    out = []
    for src in SOURCES:
        n = PARTS.get(src, 1)
        for k in range(n):
            obj = src.replace(".hip", f"_p{k}.o" if n > 1 else ".o")
            out.append((src, obj, EXTRA_FLAGS.get(src, []) + ([f"-DGSCAN_DEC_PART={k}"] if n > 1 else [])))
    return out

FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function",
         f"-I{INCLUDE}", f"-I{CSRC}"]
# per-source extras: the grouped GEMM's problem lookup reads its header from preloaded kernel-argument SGPRs
EXTRA_FLAGS = {"gemm.hip": ["-mllvm", "-amdgpu-kernarg-preload-count=12"],
               "gemm_mt.hip": ["-mllvm", "-amdgpu-kernarg-preload-count=12"]}


def _hipcc() -> str:
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: the HIP library cannot be built on this machine")
    return exe


def _stale(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False) -> str:
    os.makedirs(OBJ, exist_ok=True)
    hipcc = _hipcc()
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    headers.append(os.path.join(INCLUDE, "gscan_hip.h"))
    headers.append(os.path.abspath(__file__))
    jobs = []
    for src, obj, extra in units():
        s = os.path.join(CSRC, src)
        o = os.path.join(OBJ, obj)
        if force or _stale(o, [s] + headers):
            jobs.append([hipcc, *FLAGS, *extra, "-c", s, "-o", o])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed:\n{' '.join(cmd)}\n{r.stdout}\n{r.stderr}")
        if verbose and r.stderr.strip():
            print(r.stderr, flush=True)

    with ThreadPoolExecutor(max_workers=6) as pool:
        list(pool.map(run, jobs))
    objs = [os.path.join(OBJ, obj) for _, obj, _ in units()]
    if force or jobs or _stale(LIB, objs):
        run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", LIB])
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose="--verbose" in sys.argv))
