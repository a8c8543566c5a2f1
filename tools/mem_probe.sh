#!/bin/bash
# HBM in use by the bench at several pipeline depths: 2 ranks rehearsed on ONE GPU over the shared-memory transport, and
# the plain single-GPU pipeline.  usage (on the GPU box): bash tools/mem_probe.sh
set -eu
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R"
for cfg in "2 4" "3 6"; do set -- $cfg
SCALCE_COMM=shm SCALCE_BENCH_BUCKET_SET=1500000000 timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29534 bench.py --gpus 2 --steps 6 --warmup 2 --reads 20000000 --cpu-sample 0 --no-e2e --group $1 --inflight $2 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('world2', sys.argv[1:], 'hbm', d['config']['hbm_used_gb'], 'ms', d['ms_per_step'])" $cfg
done
timeout -k 10 300 python bench.py --steps 6 --warmup 2 --reads 20000000 --cpu-sample 0 --no-e2e --group 3 --inflight 6 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('world1 3 6 hbm', d['config']['hbm_used_gb'])"
