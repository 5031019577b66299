#!/bin/bash
# A/B of library variants on the GPU box: per-family kernel times of a short bench run.
#   bash tools/ab_families.sh <out.txt> lib1.so lib2.so ...   ("base" = the in-tree library)
out=$1; shift
mkdir -p "$(dirname "$out")"; : > "$out"
for rep in 1 2; do for lib in "$@"; do
  if [ "$lib" = base ]; then unset GSCAN_HIP_LIB; else export GSCAN_HIP_LIB=$lib; fi
  echo "== $lib (rep $rep)" >> "$out"
  timeout -k 10 200 python bench.py --cpu-seconds 0 --steps 100 --warmup 20 --warmup-seconds 1 --windows 2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); f=d['kernel_families']
print('ms/step', d['ms_per_step'], 'first', d['ms_per_step_first_window'], ' '.join(f'{k}={v[\"avg_us\"]:.1f}' for k,v in f.items()))" >> "$out"
done; done
cat "$out"
