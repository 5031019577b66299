#!/bin/bash
# A/B of an environment knob on the GPU box: bash tools/ab.sh VAR a b  -> bench lines (ms_per_step) for VAR=a and VAR=b, twice each
var=$1; shift
for rep in 1 2; do for v in "$@"; do
  echo -n "$var=$v: "; env $var=$v timeout -k 10 200 python bench.py --cpu-seconds 0 --steps 100 --warmup 20 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], d['roofline']['avg_launch_us'])"
done; done
