#!/bin/bash
# A/B of an environment knob on the GPU box: per-family kernel times of a short bench run, two repetitions each.
#   bash tools/ab_env.sh <out.txt> VAR=a VAR=b ...      (an argument "-" = the default environment)
out=$1; shift
: > "$out"
for rep in 1 2; do for kv in "$@"; do
  echo "== $kv (rep $rep)" >> "$out"
  ( [ "$kv" != "-" ] && export "$kv"; timeout -k 10 200 python bench.py --cpu-seconds 0 --steps 100 --warmup 20 --warmup-seconds 1 --windows 2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); f=d['kernel_families']
print('ms/step', d['ms_per_step'], 'first', d['ms_per_step_first_window'], ' '.join(f'{k}={v[\"avg_us\"]:.1f}' for k,v in f.items()))" ) >> "$out"
done; done
cat "$out"
