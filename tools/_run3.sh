set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/g3
timeout -k 10 900 python -m pytest tests -m gpu -q > gpurun_out/g3/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -15 gpurun_out/g3/pytest_gpu.log
timeout -k 10 300 python bench.py --cpu-seconds 0 --steps 40 > gpurun_out/g3/bench.json 2> gpurun_out/g3/bench.err || tail -5 gpurun_out/g3/bench.err
python - <<'PY'
import json
d=json.load(open("gpurun_out/g3/bench.json"))
print(d["ms_per_step"], d["ms_per_step_windows"], d["final_loss"])
for k,v in d["kernel_families"].items(): print(k, v["ms_per_step"], v["avg_us"])
PY
