set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/g5
GSCAN_DEC16=1 timeout -k 10 900 python -m pytest tests/test_parity_gpu.py tests/test_full_size_gpu.py tests/test_properties_gpu.py -m gpu -q -x > gpurun_out/g5/pytest16.log 2>&1; echo "pytest rc=$?"; tail -12 gpurun_out/g5/pytest16.log
for v in 0 1; do
  echo "== GSCAN_DEC16=$v stamps"
  GSCAN_DEC16=$v python tools/decoder_stamps.py 2>&1 | grep -v amdgpu.ids | head -5
  echo "== GSCAN_DEC16=$v bench"
  GSCAN_DEC16=$v timeout -k 10 300 python bench.py --cpu-seconds 0 --steps 40 > gpurun_out/g5/bench$v.json 2> gpurun_out/g5/bench$v.err || tail -5 gpurun_out/g5/bench$v.err
  python -c "
import json; d=json.load(open('gpurun_out/g5/bench$v.json')); k=d['kernel_families']
print(d['ms_per_step'], d['ms_per_step_windows'], 'dec fwd', k['decoder_forward']['avg_us'], 'bwd', k['decoder_backward']['avg_us'])"
done
