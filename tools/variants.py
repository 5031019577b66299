"""Development aid: build experimental copies of libgscan_hip.so that differ in -D flags for one source file.

    python tools/variants.py name1:decoder.hip:-DFOO=1 name2:gemm.hip+capi.hip:-DBAR=2,-DBAZ name3:all:-DQUX ...

Each variant relinks the objects of the normal build with the one recompiled source and lands in
variants/libgscan_hip.<name>.so (git-ignored; it travels with gpurun).  Select one at run time with
GSCAN_HIP_LIB=variants/libgscan_hip.<name>.so.
"""
import os
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multimodal_seq2seq_gscan_amd import build as B

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "variants")


def main():
    B.build()
    os.makedirs(OUT, exist_ok=True)
    hipcc = B._hipcc()
    procs = []
    for spec in sys.argv[1:]:
        name, src, flags = (spec.split(":") + [""])[:3]
        srcs = B.SOURCES if src == "all" else src.split("+")   # "all": every source with the flags; a+b: those sources
        jobs = []
        for one, unit_obj, extra in B.units():                  # (decoder.hip is four translation units)
            if one not in srcs:
                continue
            obj = os.path.join(OUT, f"{name}.{unit_obj}")
            cmd = [hipcc, *B.FLAGS, *extra, *[f for f in flags.split(",") if f], "-c", os.path.join(B.CSRC, one), "-o", obj]
            jobs.append((unit_obj, obj, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)))
        procs.append((name, jobs))
    for name, jobs in procs:
        built = {}
        for unit_obj, obj, p in jobs:
            out, err = p.communicate()
            if p.returncode != 0:
                raise SystemExit(f"{name}: hipcc failed\n{err}")
            built[unit_obj] = obj
        objs = [built.get(u, os.path.join(B.OBJ, u)) for _, u, _ in B.units()]
        lib = os.path.join(OUT, f"libgscan_hip.{name}.so")
        subprocess.run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", lib], check=True)
        print(lib)


if __name__ == "__main__":
    main()
