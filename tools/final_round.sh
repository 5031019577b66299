#!/bin/bash
# The extras of an evidence round, on the GPU box AFTER `bash tools/gpu_round.sh <tag>` and after that run's pmc_traffic.json was copied to
# profiles/pmc_traffic_latest.json (the bench lines then carry traffic bound to the kernel sources' hash): S1 / S3 / S4 lines, smoke,
# one-rank data-parallel overheads, the batcher leg, one-rank RCCL lines.   bash tools/final_round.sh <tag>  -> gpurun_out/<tag>/
set -o pipefail
tag=${1:-r06_y}; out=gpurun_out/$tag; mkdir -p $out
timeout -k 10 300 python bench.py > $out/bench_line.json 2> $out/bench.err || { tail -20 $out/bench.err; exit 1; }
python -c "import json; d=json.load(open('$out/bench_line.json')); print(d['value'], d['ms_per_step'], d['roofline']['traffic'], d['roofline']['traffic_source'])"
timeout -k 10 300 python bench.py --workload target_length --target-length 120 --steps 20 --warmup 5 --cpu-seconds 8 > $out/bench_S3.json 2>> $out/bench.err
timeout -k 10 300 python bench.py --auxiliary --cpu-seconds 0 > $out/bench_S4.json 2>> $out/bench.err
python __graft_entry__.py smoke > $out/smoke.txt 2>&1; tail -1 $out/smoke.txt
timeout -k 10 400 python tools/dp_overhead.py 2>&1 | grep "auxiliary=" > $out/dp_overhead.txt; cat $out/dp_overhead.txt
timeout -k 10 400 python bench.py --with-batcher --cpu-seconds 0 2>/dev/null > $out/bench_with_batcher.json; python -c "import json; d=json.load(open('$out/bench_with_batcher.json')); print(d['ms_per_step'], json.dumps(d['with_batcher']['reference_order']), json.dumps(d['with_batcher']['length_buckets_8']))"
WORLD_SIZE=1 RANK=0 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29611 timeout -k 10 300 python bench.py --always-collective --cpu-seconds 0 2>/dev/null > $out/bench_one_rank_rccl.json; python -c "import json; d=json.load(open('$out/bench_one_rank_rccl.json')); print('one-rank rccl', d['ms_per_step'], d['config']['rccl_nranks'], d['config']['gradient_exchange'])"
GSCAN_DP_BUCKETS=2 WORLD_SIZE=1 RANK=0 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29612 timeout -k 10 300 python bench.py --always-collective --cpu-seconds 0 2>/dev/null > $out/bench_one_rank_rccl_two_buckets.json; python -c "import json; d=json.load(open('$out/bench_one_rank_rccl_two_buckets.json')); print('two buckets', d['ms_per_step'], d['config']['dp_buckets'])"
