"""Host-only sanitizer build of the library's planners (sanitizers; GPU AddressSanitizer is not available on
this pool, and is not what host code needs).

Every source of libgscan_hip.so is compiled with `hipcc --offload-host-only -DGSCAN_PLAN_ONLY
-fsanitize=address,undefined` — the host half only: no device code is generated, device calls become no-ops and kernel
launches are checked and counted (csrc/common.h) — and linked with tests/planner/driver.hip into `planner_asan`.  The
objects reference their (absent) device code object through one `__hip_fatbin_<id>` symbol each; a generated C file defines
those as empty blobs, which the HIP runtime registers lazily and never opens because nothing is launched.

    python tests/planner/build.py          -> tests/planner/build/planner_asan
"""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from multimodal_seq2seq_gscan_amd import build as product   # noqa: E402  (the product's own source list and hipcc)

OUT = os.path.join(HERE, "build")
BINARY = os.path.join(OUT, "planner_asan")
FLAGS = ["--offload-host-only", "-DGSCAN_PLAN_ONLY", "-O1", "-g", "-std=c++17", "-fno-omit-frame-pointer",
         "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-Wno-unused-value", "-Wno-unused-variable",
         "-Wno-unused-but-set-variable", "-Wno-unused-function", f"-I{product.INCLUDE}", f"-I{product.CSRC}"]


def build(force: bool = False) -> str:
    os.makedirs(OUT, exist_ok=True)
    hipcc = product._hipcc()
    units = [(os.path.join(product.CSRC, s), o, [f for f in extra if f.startswith("-D")]) for s, o, extra in product.units()]
    units.append((os.path.join(HERE, "driver.hip"), "driver.o", []))
    sources = [u[0] for u in units]
    deps = sources + [os.path.join(product.CSRC, f) for f in os.listdir(product.CSRC) if f.endswith(".h")] + \
        [os.path.join(product.INCLUDE, "gscan_hip.h"), os.path.abspath(__file__)]
    if not force and os.path.exists(BINARY) and all(os.path.getmtime(d) <= os.path.getmtime(BINARY) for d in deps):
        return BINARY

    def compile_one(unit) -> str:
        src, name, defines = unit
        obj = os.path.join(OUT, name)
        r = subprocess.run([hipcc, *FLAGS, *defines, "-c", src, "-o", obj], capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"host-only build of {src} failed:\n{r.stdout}\n{r.stderr}")
        return obj

    with ThreadPoolExecutor(max_workers=4) as pool:
        objs = list(pool.map(compile_one, units))
    # the device code objects these host halves would register: empty stand-ins, one per translation unit
    nm = subprocess.run(["nm", *objs], capture_output=True, text=True, check=True).stdout
    fat = sorted({line.split()[-1] for line in nm.splitlines() if " U __hip_fatbin_" in line})
    stub = os.path.join(OUT, "fatbin_stubs.c")
    with open(stub, "w") as f:
        for sym in fat:
            f.write(f"const char {sym}[64] __attribute__((aligned(4096))) = {{0}};\n")
    stub_o = stub.replace(".c", ".o")
    subprocess.run(["gcc", "-c", stub, "-o", stub_o], check=True)
    r = subprocess.run([hipcc, "--offload-host-only", "-fsanitize=address,undefined", *objs, stub_o, "-ldl", "-o", BINARY],
                       capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link of planner_asan failed:\n{r.stdout}\n{r.stderr}")
    return BINARY


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
