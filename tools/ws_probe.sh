#!/bin/bash
# rocprofv3 kernel statistics of the bench with the weights-stationary GEMM on: average duration of the forward launch
export TMPDIR=/tmp
out=${1:-gpurun_out/ws_prof}
rm -rf "$out"; mkdir -p "$out"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$out" -o run -- python3 bench.py --steps 50 --warmup 10 --warmup-seconds 0.3 --windows 1 --cpu-seconds 0 > "$out/bench.log" 2>&1
f=$(find "$out" -name '*kernel_stats.csv' | head -1)
python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"]
    if "gemm" in n or "decoder" in n:
        print(f'{n[:60]:60s} calls {r["Calls"]:>6s} avg {float(r["AverageNs"])/1e3:8.1f} us  min {float(r["MinNs"])/1e3:8.1f}  max {float(r["MaxNs"])/1e3:8.1f}')
PY
rm -rf "$out"/*/ 2>/dev/null
