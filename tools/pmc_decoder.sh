#!/bin/bash
# SQ instruction / busy / wait counters of the decoder kernels (development aid): three rocprofv3 PMC passes over a short
# bench run, per-kernel averages printed with tools/pmc_by_launch.py.   usage: bash tools/pmc_decoder.sh <tag> [kernel]
tag=${1:-pmc}; kernel=${2:-decoder_fwd_kernel}
out=gpurun_out/$tag
mkdir -p "$out"
export TMPDIR=/tmp
rocprofv3 -L > "$out/counters_avail.txt" 2>&1
run() {  # name, counters...
  name=$1; shift
  timeout -k 10 240 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d "$out/$name" -o run -- python3 bench.py --steps 5 --warmup 3 --windows 0 --cpu-seconds 0 > "$out/$name.log" 2>&1
  f=$(find "$out/$name" -name '*counter_collection.csv' | head -1)
  if [ -n "$f" ]; then python tools/pmc_by_launch.py "$f" "$kernel" | tee "$out/$name.txt"; else tail -5 "$out/$name.log"; fi
  rm -rf "$out/$name"
}
run insts SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VALU_MFMA_MOPS_F32
run active SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES
run waits SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_SALU
