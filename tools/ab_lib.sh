#!/bin/bash
# A/B of library variants on the GPU box (tools/variants.py builds them): for each library, the decoder kernels' phase
# stamps and a short bench line.   bash tools/ab_lib.sh <out.txt> lib1.so lib2.so ...   ("base" = the in-tree library)
out=$1; shift
: > "$out"
for rep in 1 2; do for lib in "$@"; do
  if [ "$lib" = base ]; then unset GSCAN_HIP_LIB; else export GSCAN_HIP_LIB=$lib; fi
  echo "== $lib (rep $rep)" >> "$out"
  if [ $rep = 1 ] && [ -z "$NOSTAMPS" ]; then timeout -k 10 200 python tools/decoder_stamps.py 2>/dev/null | grep -v amdgpu >> "$out"; fi
  timeout -k 10 200 python bench.py --cpu-seconds 0 --steps 100 --warmup 20 --warmup-seconds 1 --windows 2 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); f=d['kernel_families']; print('ms/step', d['ms_per_step'], 'first', d['ms_per_step_first_window'], 'dec fwd/bwd us', f['decoder_forward']['avg_us'], f['decoder_backward']['avg_us'], 'gemm ms', f['gemm']['ms_per_step'])" >> "$out"
done; done
cat "$out"
