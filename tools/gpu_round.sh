#!/bin/bash
# One GPU call's worth of evidence: parity tests, bench line, rocprofv3 kernel stats, PMC traffic passes.
# usage (on the GPU box, from the repo root): bash tools/gpu_round.sh <tag>   -> files under gpurun_out/<tag>/
# (then copy <tag>/pmc_traffic.json to profiles/pmc_traffic_latest.json and run tools/final_round.sh <tag> for the bound bench lines)
set -eo pipefail
tag=${1:-run}
out=gpurun_out/$tag
mkdir -p "$out"
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests -m gpu -x -q > "$out/pytest_gpu.log" 2>&1 || { tail -30 "$out/pytest_gpu.log"; exit 1; }
tail -3 "$out/pytest_gpu.log"
timeout -k 10 300 python bench.py > "$out/bench_line.json" 2> "$out/bench.err" || { tail -30 "$out/bench.err"; exit 1; }
cat "$out/bench_line.json"
# the other single-GPU configurations of BASELINE.json: S3 (k=13, T=120) and S4 (auxiliary head)
timeout -k 10 300 python bench.py --workload target_length --target-length 120 --steps 20 --warmup 5 --cpu-seconds 8 > "$out/bench_S3.json" 2>> "$out/bench.err" || { tail -30 "$out/bench.err"; exit 1; }
timeout -k 10 300 python bench.py --auxiliary --cpu-seconds 0 > "$out/bench_S4.json" 2>> "$out/bench.err" || { tail -30 "$out/bench.err"; exit 1; }
python -c "import json,sys; [print(f, json.load(open(f))['value'], json.load(open(f))['ms_per_step']) for f in sys.argv[1:]]" "$out/bench_S3.json" "$out/bench_S4.json"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/prof" -o run -- python3 bench.py --cpu-seconds 0 > "$out/prof_bench.log" 2>&1
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$out/pmc_f" -o run -- python3 bench.py --steps 20 --warmup 5 --warmup-seconds 0.1 --cpu-seconds 0 > "$out/pmc_f.log" 2>&1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$out/pmc_w" -o run -- python3 bench.py --steps 20 --warmup 5 --warmup-seconds 0.1 --cpu-seconds 0 > "$out/pmc_w.log" 2>&1
timeout -k 10 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d "$out/pmc_m" -o run -- python3 bench.py --steps 20 --warmup 5 --warmup-seconds 0.1 --cpu-seconds 0 > "$out/pmc_m.log" 2>&1
python tools/pmc_mfma.py "$(find $out/pmc_m -name '*counter_collection.csv' | head -1)" > "$out/pmc_mfma.json"
python tools/pmc_traffic.py "$(find $out/pmc_f -name '*counter_collection.csv' | head -1)" "$(find $out/pmc_w -name '*counter_collection.csv' | head -1)" all "$tag" > "$out/pmc_traffic.json"
cat "$out/pmc_traffic.json"
# world_channel_lists_kernel is launched on a side stream BEFORE the decoder's reverse kernel and queues behind its full
# register file; rocprofv3 --stats counts the wait (~100 us average) as kernel time.  Its row is rewritten from the
# serialised PMC pass's kernel trace (tools/kernel_stats_fix.py); the uncorrected table is kept as kernel_stats_raw.csv.
cp "$(find $out/prof -name '*kernel_stats.csv' | head -1)" "$out/kernel_stats_raw.csv"
python tools/kernel_stats_fix.py "$out/kernel_stats_raw.csv" "$(find $out/pmc_f -name '*kernel_trace.csv' | head -1)" > "$out/kernel_stats.csv"
head -12 "$out/kernel_stats.csv"
# timeline of an undisturbed step from in-kernel clock stamps (needs: python tools/variants.py trace:all:-DGSCAN_TRACE)
if [ -f variants/libgscan_hip.trace.so ]; then
  python tools/device_timeline.py > "$out/device_timeline_three_streams.txt" 2>&1
  python tools/device_timeline.py --single-stream > "$out/device_timeline_single_stream.txt" 2>&1
  tail -3 "$out/device_timeline_three_streams.txt"
fi
# raw traces are large: keep only the summaries
rm -rf "$out/prof" "$out/pmc_f" "$out/pmc_w" "$out/pmc_m"
