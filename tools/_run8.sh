#!/bin/bash
set -o pipefail
timeout -k 10 600 python -m pytest tests/test_parity_gpu.py -m gpu -q -k "streamed_shapes or streaming_kernels" > gpurun_out/any_greedy.log 2>&1
echo "exit $?" >> gpurun_out/any_greedy.log
tail -40 gpurun_out/any_greedy.log
timeout -k 10 900 python tools/fuzz_parity.py --extremes --train-step > gpurun_out/fuzz_extremes.log 2>&1
echo "exit $?" >> gpurun_out/fuzz_extremes.log
tail -20 gpurun_out/fuzz_extremes.log
timeout -k 10 900 python tools/fuzz_parity.py --wide --cases 60 --seed 4 --train-step > gpurun_out/fuzz_wide.log 2>&1
echo "exit $?" >> gpurun_out/fuzz_wide.log
tail -70 gpurun_out/fuzz_wide.log
