#!/bin/bash
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r5_run5; mkdir -p $O
BENCH_TRACE=1 timeout -k 10 500 python bench.py --steps 20 --warmup 5 --no-e2e --cpu-sample 0 --no-verify > $O/trace_default.txt 2>&1
BENCH_TRACE=1 SCALCE_BENCH_NO_SIDE=1 SCALCE_BENCH_NO_INPLACE=1 timeout -k 10 500 python bench.py --steps 20 --warmup 5 --no-e2e --cpu-sample 0 --no-verify > $O/trace_old.txt 2>&1
BENCH_TRACE=1 timeout -k 10 500 python bench.py --steps 20 --warmup 5 --no-e2e --cpu-sample 0 --no-verify --group 6 --inflight 18 > $O/trace_g6_18.txt 2>&1
BENCH_TRACE=1 timeout -k 10 500 python bench.py --steps 20 --warmup 5 --no-e2e --cpu-sample 0 --no-verify --group 4 --inflight 16 > $O/trace_g4_16.txt 2>&1
for f in $O/trace_*.txt; do echo $f; grep "^{" $f | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['ms_per_step'], j['config']['shards_in_flight'], j['config']['shards_per_coder_launch'], j['config']['coder_streams'], j['roofline']['kernel']['launch_ms'])"; done
