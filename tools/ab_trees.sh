#!/bin/bash
# A/B of whole source trees on the GPU box (an older commit copied under variants/<name>_tree with its own built library):
#   bash tools/ab_trees.sh <out.txt> dir1 dir2 ...     ("." = this tree); per-family kernel times, two repetitions each
out=$(realpath "$1"); shift
: > "$out"
root=$(pwd)
for rep in 1 2 3; do for dir in "$@"; do
  echo "== $dir (rep $rep)" >> "$out"
  ( cd "$root/$dir" && timeout -k 10 200 python bench.py --cpu-seconds 0 --steps 100 --warmup 20 --warmup-seconds 1 --windows 2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); f=d['kernel_families']
print('ms/step', d['ms_per_step'], 'first', d['ms_per_step_first_window'], ' '.join(f'{k}={v[\"avg_us\"]:.1f}' for k,v in f.items()))" ) >> "$out"
done; done
cat "$out"
