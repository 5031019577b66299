#!/bin/bash
# Scaling runs of the data-parallel training step on ONE node with up to 8 MI355X (BASELINE.json
# configs[2] and [4]).  Not runnable on the one-GPU development box: written so that whoever has an 8-GPU node can run it
# as is.  Every run is a set of FRESH child processes (python bench.py --gpus N starts its own ranks before anything
# touches the device; no process that has initialised HIP is ever re-exec'ed).
#
#   bash tools/scale_run.sh [out_dir]          -> out_dir/{weak_N,strong_N,aux_4}.json + out_dir/summary.txt
#
# What is run:  weak scaling 256 rows per GPU at N = 1, 2, 4, 8 (configs[1] -> configs[2]); strong scaling at a global
# batch of 2048 rows for N = 2, 4, 8 (SURVEY.md 8d's secondary line); configs[4]: the auxiliary head on 4 GPUs.
# What is checked on every line with N > 1:  config.rccl_nranks == N (the communicator the gradients travel on, as RCCL
# itself reports it: N independent replicas cannot pass), config.gradient_exchange names the library's own all-reduce on
# the step's stream, and the 8-rank weak-scaling step stays within DESIGN.md 8's budget (<= 0.57 ms).
set -eo pipefail
out=${1:-gpurun_out/scale}
mkdir -p "$out"
export HSA_ENABLE_IPC_MODE_LEGACY=0
steps=${STEPS:-50}
warmup=${WARMUP:-10}
visible=$(python3 - <<'EOF'
import os, sys
sys.path.insert(0, os.getcwd())
from bench import visible_gpu_count          # sysfs only: no HIP call in this shell's helpers either
print(visible_gpu_count() or 0)
EOF
)
echo "visible GPUs: $visible" | tee "$out/summary.txt"
run() {   # name, gpus, extra bench.py arguments...
  local name=$1 n=$2; shift 2
  if [ "$n" -gt "$visible" ]; then echo "$name: skipped ($n GPUs needed, $visible visible)" | tee -a "$out/summary.txt"; return 0; fi
  timeout -k 10 900 python3 bench.py --gpus "$n" --steps "$steps" --warmup "$warmup" --cpu-seconds 0 "$@" \
      > "$out/$name.json" 2> "$out/$name.err" || { echo "$name: FAILED (rc $?)" | tee -a "$out/summary.txt"; tail -20 "$out/$name.err"; return 1; }
}
for n in 1 2 4 8; do run "weak_$n" "$n"; done
# the two-bucket exchange (GSCAN_DP_BUCKETS=2: early gradients all-reduced under the backward tail) beside the one-bucket line
for n in 2 4 8; do GSCAN_DP_BUCKETS=2 run "weak2b_$n" "$n"; done
for n in 2 4 8; do run "strong_$n" "$n" --global-batch 2048; done
run aux_4 4 --auxiliary
python3 - "$out" <<'EOF' | tee -a "$out/summary.txt"
import glob, json, os, sys
out = sys.argv[1]
rows, bad = {}, []
for path in sorted(glob.glob(os.path.join(out, "*.json"))):
    name = os.path.basename(path)[:-5]
    try:
        d = json.load(open(path))
    except ValueError:
        bad.append(f"{name}: no JSON line")
        continue
    n, cfg = d["n_gpus"], d["config"]
    rows[name] = d
    if n > 1:
        if cfg.get("rccl_nranks") != n:
            bad.append(f"{name}: rccl_nranks {cfg.get('rccl_nranks')} != {n} (independent replicas, or the fallback transport)")
        if "gscan_allreduce_f32" not in str(cfg.get("gradient_exchange")):
            bad.append(f"{name}: gradient_exchange = {cfg.get('gradient_exchange')!r}, expected the library's RCCL all-reduce")
    w = d["ms_per_step_windows"]
    print(f"{name:10s} N={n} {d['scaling']:6s} global batch {cfg['global_batch']:5d}  median {d['ms_per_step']:.4f} ms/step  "
          f"first window {d['ms_per_step_first_window']:.4f}  windows {w['min']:.4f}-{w['max']:.4f}  {d['value']:.0f} examples/s  "
          f"rccl_nranks {cfg.get('rccl_nranks')}  buckets {cfg.get('dp_buckets')}  value = {cfg.get('value_window')}  "
          f"exchange {cfg.get('gradient_exchange')}")
if "weak_1" in rows:
    base = rows["weak_1"]["value"]
    for n in (2, 4, 8):
        if f"weak_{n}" in rows:
            print(f"weak scaling at {n}: {rows[f'weak_{n}']['value'] / base:.2f}x of one GPU (ideal {n}x; north star: >= 6x at 8)")
if "weak_8" in rows and rows["weak_8"]["ms_per_step"] > 0.57:
    bad.append(f"weak_8: {rows['weak_8']['ms_per_step']:.4f} ms per step is over DESIGN.md 8's budget of 0.57 ms")
for n in (2, 4, 8):
    if f"weak_{n}" in rows and f"weak2b_{n}" in rows:
        print(f"two buckets at {n}: {rows[f'weak2b_{n}']['ms_per_step']:.4f} vs one bucket {rows[f'weak_{n}']['ms_per_step']:.4f} ms/step")
for line in bad:
    print("CHECK FAILED:", line)
sys.exit(1 if bad else 0)
EOF
