# diagnostic: the bench with the one-block-per-lane coder, shards per launch / in flight / coder streams from the command line
set -e
export SCALCE_AC_BLOCKS_PER_WG=64
for cfg in "$@"; do
  set -- $cfg
  SCALCE_BENCH_CODER_STREAMS=$3 python bench.py --steps ${STEPS:-24} --group $1 --inflight $2 --no-e2e --no-verify --cpu-sample 0 2>gpurun_out/lb.err | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('G=$1 D=$2 S=$3 shared=${SCALCE_AC_LANES_SHARED:-0}', d['value'], 'MB/s', d['ms_per_step'], 'ms/step', 'hbm', d['config']['hbm_used_gb'], 'launch_ms', d['roofline']['kernel']['launch_ms'])
"
done
