#!/bin/bash
# round 5: the rocprofv3 summaries behind profiles/r05_final_* and the sharded path's numbers with this build
set -u
R=$(cd "$(dirname "$0")/.." && pwd)
cd $R
bash tools/profile_round.sh r05_final 2>&1 | tail -3
bash tools/pmc_sq.sh r05_final 2>&1 | tail -4
bash tools/kernel_times.sh r05_final 2>&1 | tail -2
O=gpurun_out/r5_profiles; mkdir -p $O
echo "== sharded world 1 (RCCL), 20 steps" > $O/sharded.log
SCALCE_BENCH_FORCE_SHARDED=1 timeout -k 10 500 python bench.py --steps 20 --warmup 5 --no-e2e --cpu-sample 0 2>&1 | grep -v "amdgpu.ids\|version\|Hostname\|Librccl" >> $O/sharded.log
echo "== sharded world 1, trace, one shard at a time" >> $O/sharded.log
SCALCE_TRACE=1 SCALCE_BENCH_FORCE_SHARDED=1 timeout -k 10 300 python bench.py --group 1 --inflight 1 --steps 2 --warmup 1 --no-e2e --no-verify --cpu-sample 0 2>&1 | grep "rank 0" | tail -8 >> $O/sharded.log
echo "== world 2 over shm, two fronts, 10 M reads per rank" >> $O/sharded.log
SCALCE_COMM=shm SCALCE_BENCH_BUCKET_SET=800000000 timeout -k 10 500 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29534 bench.py --gpus 2 --steps 8 --warmup 2 --reads 10000000 --no-e2e --cpu-sample 300000 2>&1 | grep -v "Gloo\|socket.cpp\|amdgpu.ids\|^\*\*\*\|OMP_NUM" >> $O/sharded.log
echo "== table scale" > $O/table_scale.log
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_ts -o ts -- python3 $R/tools/table_scale_probe.py 2000000 8000000 >> $R/$O/table_scale.log 2>&1
cd $R
python3 tools/prof_summary.py stats gpurun_out/prof_ts gpurun_out/r05_table_scale_kernel_stats.csv 2>&1 | tail -1
rm -rf gpurun_out/prof_ts
grep "^{" $O/table_scale.log
