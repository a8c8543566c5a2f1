#!/bin/bash
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r5_run4; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_parity.py -x -q -m gpu > $O/tests.txt 2>&1; echo "tests rc $?" >> $O/tests.txt
tail -3 $O/tests.txt
run() { # label, env...
  echo "== $1" >> $O/log.txt; shift
  env "$@" timeout -k 10 500 python bench.py --steps 20 --warmup 5 --no-e2e --cpu-sample 0 >> $O/log.txt 2>&1
}
: > $O/log.txt
run "plain default (in place + side stream)" A=1
run "no side stream" SCALCE_BENCH_NO_SIDE=1
run "no in-place" SCALCE_BENCH_NO_INPLACE=1
run "neither" SCALCE_BENCH_NO_INPLACE=1 SCALCE_BENCH_NO_SIDE=1
echo done >> $O/log.txt
python - <<'P'
import json,re
for line in open('gpurun_out/r5_run4/log.txt'):
    if line.startswith('=='): print(line.strip())
    if line.startswith('bench:'): print('  ', line.strip())
    if line.startswith('{'):
        j=json.loads(line); c=j['config']
        print('   ms_per_step %.2f value %.0f parity %s slots %d G %d streams %d hbm %.0f single %.0f launch_ms %.0f' % (j['ms_per_step'], j['value'], j['parity_checked'], c['shards_in_flight'], c['shards_per_coder_launch'], c['coder_streams'], c['hbm_used_gb'], c['ms_single_shard_alone'], j['roofline']['kernel']['launch_ms']))
P
