set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/g6
GSCAN_DEC16=1 timeout -k 10 900 python -m pytest tests/test_parity_gpu.py tests/test_full_size_gpu.py -m gpu -q -x -k "compositional or geca or target_length or full or S1 or S3 or S4 or ragged" > gpurun_out/g6/pytest16.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/g6/pytest16.log
echo "== GSCAN_DEC16=1 stamps"; GSCAN_DEC16=1 python tools/decoder_stamps.py 2>&1 | grep -v amdgpu.ids | head -5
echo "== 8 waves, SQ counters"; bash tools/pmc_decoder.sh g6/pmc8 decoder_fwd_kernel 2>&1 | grep -v amdgpu.ids | tail -12
echo "== 16 waves, SQ counters"; GSCAN_DEC16=1 bash tools/pmc_decoder.sh g6/pmc16 decoder_fwd16_kernel 2>&1 | grep -v amdgpu.ids | tail -12
