set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/g4
timeout -k 10 900 python -m pytest tests/test_properties_gpu.py tests/test_data_parallel_gpu.py -m gpu -q -k "fused_prologue or nccl_path or strong_scaling" > gpurun_out/g4/pytest.log 2>&1; echo "pytest rc=$?"; tail -25 gpurun_out/g4/pytest.log
for cfg in "GSCAN_FUSED_ACQUIRE=0" "GSCAN_FUSED_ACQUIRE=1" "GSCAN_FUSED_PROLOGUE=0" "GSCAN_FUSED_ACQUIRE=0" "GSCAN_FUSED_ACQUIRE=1" "GSCAN_FUSED_PROLOGUE=0"; do
  echo "== $cfg"
  env $cfg timeout -k 10 300 python bench.py --cpu-seconds 0 --steps 40 > gpurun_out/g4/bench.json 2> gpurun_out/g4/bench.err || tail -5 gpurun_out/g4/bench.err
  python -c "import json; d=json.load(open('gpurun_out/g4/bench.json')); print(d['ms_per_step'], d['ms_per_step_windows'])"
done
