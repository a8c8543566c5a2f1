#!/bin/bash
# Where the front stages of ONE shard alone spend wall time: kernel trace of `bench.py --group 1 --inflight 1`, then per shard the
# span from the first front-stage kernel to the coder's start, the sum of kernel times in it, and the largest idle gaps.
#   bash tools/front_gaps.sh   -> gpurun_out/front_gaps.txt
set -eu
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/prof_gaps; rm -rf $O; mkdir -p $O
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O -o g -- python3 $R/bench.py --steps 3 --warmup 1 --group 1 --inflight 1 --cpu-sample 0 --no-e2e --no-verify > $O/log.txt 2>&1
python3 - $O > $R/gpurun_out/front_gaps.txt <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
rows = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].split('(')[0].replace('void ', '').replace('scalce::', '')[:40]) for r in csv.DictReader(open(f))]
rows.sort()
# shards: split at ingest (index_count_k) starts
starts = [i for i, r in enumerate(rows) if r[2].startswith('index_count_k')]
for si, a in enumerate(starts):
    b = starts[si + 1] if si + 1 < len(starts) else len(rows)
    seg = [r for r in rows[a:b] if not r[2].startswith('ac_encode') and not r[2].startswith('ac_frame')]
    if not seg: continue
    t0, t1 = seg[0][0], max(r[1] for r in seg)
    busy = 0; cur_end = t0; gaps = []
    for s, e, n in seg:
        if s > cur_end: gaps.append((s - cur_end, n))
        busy += max(0, e - max(s, cur_end)); cur_end = max(cur_end, e)
    print("shard %d: span %.2f ms, kernels busy %.2f ms, idle %.2f ms in %d gaps, %d launches" % (si, (t1 - t0) / 1e6, busy / 1e6, (t1 - t0 - busy) / 1e6, len(gaps), len(seg)))
    gaps.sort(reverse=True)
    print("   largest gaps (us, kernel behind the gap):", [(round(g / 1e3, 1), n) for g, n in gaps[:14]])
    small = sum(g for g, n in gaps if g < 20000)
    print("   gaps below 20 us: %.2f ms in all" % (small / 1e6))
PY
cat $R/gpurun_out/front_gaps.txt
