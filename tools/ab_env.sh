#!/bin/bash
# A/B of environment knobs on the GPU box: per-family kernel times of a short bench run, two repetitions each.
#   bash tools/ab_env.sh <out.txt> VAR=a VAR=b,OTHER=c ...   (an argument "-" = the default environment; commas join knobs)
#   AB_BENCH_ARGS="--workload target_length --target-length 120" AB_STEPS=30 bash tools/ab_env.sh ...   (another workload)
out=$1; shift
mkdir -p "$(dirname "$out")"; : > "$out"
steps=${AB_STEPS:-100}
for rep in 1 2; do for kv in "$@"; do
  echo "== $kv (rep $rep)" >> "$out"
  ( if [ "$kv" != "-" ]; then IFS=, read -ra kvs <<< "$kv"; for one in "${kvs[@]}"; do export "$one"; done; fi
    timeout -k 10 200 python bench.py $AB_BENCH_ARGS --cpu-seconds 0 --steps $steps --warmup 20 --warmup-seconds 1 --windows 2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); f=d['kernel_families']
print('ms/step', d['ms_per_step'], 'first', d['ms_per_step_first_window'], ' '.join(f'{k}={v[\"avg_us\"]:.1f}' for k,v in f.items()))" ) >> "$out"
done; done
cat "$out"
