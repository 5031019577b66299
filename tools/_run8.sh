#!/bin/bash
set -o pipefail
timeout -k 10 900 python -m pytest tests/test_parity_gpu.py -m gpu -q -k "edge_shapes" > gpurun_out/any_edges.log 2>&1
echo "exit $?" >> gpurun_out/any_edges.log
tail -60 gpurun_out/any_edges.log
