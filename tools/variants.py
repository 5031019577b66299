"""Development aid: build experimental copies of libgscan_hip.so that differ in -D flags for one source file.

    python tools/variants.py name1:decoder.hip:-DFOO=1 name2:gemm.hip:-DBAR=2,-DBAZ ...

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
        obj = os.path.join(OUT, f"{name}.{src.replace('.hip', '.o')}")
        cmd = [hipcc, *B.FLAGS, *[f for f in flags.split(",") if f], "-c", os.path.join(B.CSRC, src), "-o", obj]
        procs.append((name, src, obj, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)))
    for name, src, obj, p in procs:
        out, err = p.communicate()
        if p.returncode != 0:
            raise SystemExit(f"{name}: hipcc failed\n{err}")
        objs = [obj if s == src else os.path.join(B.OBJ, s.replace(".hip", ".o")) for s in B.SOURCES]
        lib = os.path.join(OUT, f"libgscan_hip.{name}.so")
        subprocess.run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", lib], check=True)
        print(lib)


if __name__ == "__main__":
    main()
