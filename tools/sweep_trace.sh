#!/bin/bash
# Per-sweep kernel times of the tie-break of ONE shard alone: rocprofv3 kernel trace of `bench.py --group 1 --inflight 1`,
# then the durations of jacobi_k / seg_rescan_k of the last shard in dispatch order.
# usage: bash tools/sweep_trace.sh TAG [env assignments...]   -> gpurun_out/TAG_sweeps.txt
set -eu
TAG=${1:-sweeps}; shift || true
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/prof_$TAG
rm -rf $O && mkdir -p $O
for kv in "$@"; do export "$kv"; done
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/tr -o tr -- python3 $R/bench.py --steps 1 --warmup 1 --group 1 --inflight 1 --cpu-sample 0 --no-e2e --no-verify > $O/trace.log 2>&1
cd $R
python3 - "$O/tr" > gpurun_out/${TAG}_sweeps.txt <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last shard = everything behind the last ingest_tiles_k
last = max(i for i, r in enumerate(rows) if "ingest_tiles_k" in r["Kernel_Name"])
j, s = [], []
for r in rows[last:]:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    if "jacobi_k" in r["Kernel_Name"]: j.append(d)
    if "seg_rescan_k" in r["Kernel_Name"]: s.append(d)
print("jacobi_k us:", " ".join("%.0f" % x for x in j))
print("seg_rescan_k us:", " ".join("%.0f" % x for x in s))
print("sum ms: jacobi %.2f rescan %.2f" % (sum(j) / 1e3, sum(s) / 1e3))
t0 = int(rows[last]["Start_Timestamp"]); t1 = max(int(r["End_Timestamp"]) for r in rows[last:] if "ac_encode" not in r["Kernel_Name"])
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows[last:] if "ac_encode" not in r["Kernel_Name"])
print("front stages of the shard: %.1f ms from first to last kernel, %.1f ms of kernels, %d launches" % ((t1 - t0) / 1e6, busy / 1e6, len([r for r in rows[last:] if "ac_encode" not in r["Kernel_Name"]])))
PY
rm -rf $O/tr
cat gpurun_out/${TAG}_sweeps.txt
