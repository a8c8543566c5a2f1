#!/bin/bash
# The rocprofv3 runs behind profiles/<tag>_*: kernel trace + stats of the default bench, then FETCH_SIZE and WRITE_SIZE in
# separate --pmc passes (never together with a trace domain other than the kernel trace).  Run on the GPU box:
#   bash tools/profile_round.sh r01_v13
set -eu
TAG=${1:-r01_vX}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/prof_$TAG
rm -rf $O && mkdir -p $O
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o st -- python3 $R/bench.py --steps 9 --warmup 3 --cpu-sample 0 --no-e2e --no-verify > $O/stats.log 2>&1
timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch -o f -- python3 $R/bench.py --steps 3 --warmup 3 --cpu-sample 0 --no-e2e --no-verify > $O/fetch.log 2>&1
timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write -o w -- python3 $R/bench.py --steps 3 --warmup 3 --cpu-sample 0 --no-e2e --no-verify > $O/write.log 2>&1
cd $R
python3 tools/prof_summary.py stats $O/stats $R/gpurun_out/${TAG}_bench50m_kernel_stats.csv
python3 tools/prof_summary.py pmc $O/fetch $O/write $R/gpurun_out/${TAG}_bench50m_pmc_fetch_write.json
tail -1 $O/stats.log | cut -c1-300
rm -rf $O/stats $O/fetch $O/write   # the raw traces are large; the summaries are what travels back
