#!/bin/bash
# bench.py over pipeline shapes: "G D S [env...]" = shards per coder launch, shards in flight, coder streams
# usage: bash tools/pipe_shapes.sh "4 12 2" "3 12 3" ...   -> one line per shape
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
for cfg in "$@"; do
  set -- $cfg
  G=$1; D=$2; S=$3; shift 3
  env GPU_MAX_HW_QUEUES=${GPU_MAX_HW_QUEUES:-8} SCALCE_BENCH_CODER_STREAMS=$S "$@" python bench.py --group $G --inflight $D --steps ${STEPS:-36} --no-e2e --cpu-sample 0 --no-verify > gpurun_out/shape.json 2> gpurun_out/shape.err || { echo "$cfg: failed"; tail -3 gpurun_out/shape.err; continue; }
  python -c "import json; d=json.load(open('gpurun_out/shape.json')); print('$cfg:', d['ms_per_step'], 'ms/step', d['value'], 'MB/s', 'launch', d['roofline']['kernel']['launch_ms'], 'hbm', d['config']['hbm_used_gb'])"
done
