#!/bin/bash
# 2 ranks on ONE GPU over gloo: exercises bench.py's multi-rank path end to end (sharding,
# batched vocabulary exchange, moments all-reduce) and checks the merged vocabularies.
export NVT_BENCH_SHARE_GPU=1 NVT_BENCH_BACKEND=gloo HSA_ENABLE_IPC_MODE_LEGACY=0
cd /root/repo
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 2 --steps 2 --warmup 1 --rows ${1:-3000000} --no-cpu-baseline 2>&1 | tail -5
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29612 tests/multirank_check.py 2>&1 | tail -8
