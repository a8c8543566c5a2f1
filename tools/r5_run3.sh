#!/bin/bash
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r5_run3; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_sharded_cpp.py tests/test_gpu_cli.py tests/test_ref_files.py tests/test_gpu_fuzz.py -x -q -m gpu > $O/tests.txt 2>&1; echo "tests rc $?" >> $O/tests.txt
tail -3 $O/tests.txt
echo "== sharded world 1, trace, one shard at a time" > $O/log.txt
SCALCE_SHARD_TRACE=1 SCALCE_BENCH_FORCE_SHARDED=1 timeout -k 10 300 python bench.py --group 1 --inflight 1 --steps 2 --warmup 1 --no-e2e --no-verify --cpu-sample 0 2>&1 | grep -v "^{" | tail -9 >> $O/log.txt
echo "== sharded world 1, default shape, 20 steps" >> $O/log.txt
SCALCE_BENCH_FORCE_SHARDED=1 timeout -k 10 500 python bench.py --steps 20 --warmup 5 --no-e2e --cpu-sample 0 >> $O/log.txt 2>&1
echo "== world 2 over shm, 10 M reads per rank, trace" >> $O/log.txt
SCALCE_SHARD_TRACE=1 SCALCE_COMM=shm SCALCE_BENCH_BUCKET_SET=800000000 timeout -k 10 500 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29534 bench.py --gpus 2 --steps 4 --warmup 2 --reads 10000000 --no-e2e --cpu-sample 0 --group 1 --inflight 1 2>&1 | grep -v "Gloo\|socket.cpp\|amdgpu.ids" | tail -40 >> $O/log.txt
echo "== rccl big send" >> $O/log.txt
timeout -k 10 300 python tools/rccl_big_send.py 2>&1 | grep -v "amdgpu.ids" >> $O/log.txt
echo done >> $O/log.txt
