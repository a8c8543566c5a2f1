#!/bin/bash
# round 5 baseline: where the sharded path's time goes at world 1 (RCCL), and the world-2 shm rehearsal of bench.py
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r5_base; mkdir -p $O
echo "== sharded world 1, trace, one shard at a time" > $O/log.txt
SCALCE_SHARD_TRACE=1 SCALCE_BENCH_FORCE_SHARDED=1 timeout -k 10 300 python bench.py --group 1 --inflight 1 --steps 3 --warmup 1 --no-e2e --no-verify --cpu-sample 0 >> $O/log.txt 2>&1
echo "== sharded world 1, default shape, 20 steps" >> $O/log.txt
SCALCE_BENCH_FORCE_SHARDED=1 timeout -k 10 400 python bench.py --steps 20 --warmup 5 --no-e2e --cpu-sample 0 >> $O/log.txt 2>&1
echo "== world 2 over shm, 20 M reads per rank" >> $O/log.txt
SCALCE_COMM=shm SCALCE_BENCH_BUCKET_SET=1500000000 timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29534 bench.py --gpus 2 --steps 6 --warmup 2 --reads 20000000 --no-e2e >> $O/log.txt 2>&1
echo done >> $O/log.txt
