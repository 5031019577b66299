set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/g7
timeout -k 10 1000 python -m pytest tests -m gpu -q > gpurun_out/g7/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -8 gpurun_out/g7/pytest_gpu.log
timeout -k 10 300 python bench.py --cpu-seconds 0 --steps 40 > gpurun_out/g7/bench.json 2> gpurun_out/g7/bench.err || tail -5 gpurun_out/g7/bench.err
python -c "import json; d=json.load(open('gpurun_out/g7/bench.json')); print(d['ms_per_step'], d['ms_per_step_windows'], d['roofline']['kernel'], d['roofline']['frac'], d['roofline_gemm']['frac'] if d['roofline_gemm'] else None)"
GSCAN_DETERMINISTIC=1 timeout -k 10 300 python bench.py --cpu-seconds 0 --steps 40 > gpurun_out/g7/bench_det.json 2> gpurun_out/g7/bench_det.err || tail -5 gpurun_out/g7/bench_det.err
python -c "import json; d=json.load(open('gpurun_out/g7/bench_det.json')); print('deterministic', d['ms_per_step'], d['ms_per_step_windows'])"
