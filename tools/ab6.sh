#!/bin/bash
# Round 6 A/B of library variants on the GPU box (tools/variants.py builds them).  For each NAME:
#   variants/libgscan_hip.NAME.so      -> a short bench line (HIP-event kernel times), twice
#   variants/libgscan_hip.NAME_st.so   -> if present: the decoder kernels' phase stamps (a -DGSCAN_DEC_STAMPS build of the
#                                         same switches: the shipped kernels carry no stamps since round 6)
# "base" = the in-tree library.     bash tools/ab6.sh <out.txt> NAME ...        (env S3=1: the long-target workload too)
out=$1; shift
: > "$out"
bench() {
  timeout -k 10 200 python bench.py --cpu-seconds 0 --steps 100 --warmup 20 --warmup-seconds 1 --windows 3 "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); f=d['kernel_families']; print('ms/step', d['ms_per_step'], 'first', d['ms_per_step_first_window'], 'dec fwd/bwd us', f['decoder_forward']['avg_us'], f['decoder_backward']['avg_us'], 'gemm ms', f['gemm']['ms_per_step'], 'keys', f.get('keys_backward', {}).get('avg_us'))"
}
for rep in 1 2; do for name in "$@"; do
  lib=variants/libgscan_hip.$name.so; st=variants/libgscan_hip.${name}_st.so
  echo "== $name (rep $rep)" >> "$out"
  if [ $rep = 1 ] && [ -f "$st" ]; then GSCAN_HIP_LIB=$st timeout -k 10 200 python tools/decoder_stamps.py 2>/dev/null | grep -v amdgpu >> "$out"; fi
  if [ "$name" = base ]; then unset GSCAN_HIP_LIB; else export GSCAN_HIP_LIB=$lib; fi
  bench >> "$out"
  if [ -n "$S3" ] && [ $rep = 1 ]; then echo -n "S3: " >> "$out"; bench --workload target_length --target-length 120 --steps 40 >> "$out"; fi
  unset GSCAN_HIP_LIB
done; done
cat "$out"
