#!/bin/bash
# Kernel times of ONE shard alone (nothing beside it): rocprofv3 kernel trace of `bench.py --group 1 --inflight 1`.
# usage: bash tools/kernel_times.sh TAG [env assignments...]   -> gpurun_out/TAG_alone_kernel_stats.csv
set -eu
TAG=${1:-alone}; shift || true
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/prof_$TAG
rm -rf $O && mkdir -p $O
for kv in "$@"; do export "$kv"; done
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o st -- python3 $R/bench.py --steps 4 --warmup 1 --group 1 --inflight 1 --cpu-sample 0 --no-e2e --no-verify > $O/stats.log 2>&1
cd $R
python3 tools/prof_summary.py stats $O/stats $R/gpurun_out/${TAG}_alone_kernel_stats.csv
rm -rf $O/stats
