#!/bin/bash
# Instruction and stall counters of the front-stage kernels of ONE shard alone (rocprofv3 --pmc, one pass per counter group).
#   bash tools/pmc_front.sh TAG [kernel-name substring ...]   -> gpurun_out/TAG_pmc_front.txt
set -eu
TAG=${1:-front}; shift || true
PAT=${*:-ingest tokenize trigram gather_rows emit_reads tie_candidates radix_scatter_kv}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
OUT=$R/gpurun_out/${TAG}_pmc_front.txt
: > $OUT
for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU" "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD"; do
  rm -rf $R/gpurun_out/pmc_front
  timeout -k 10 300 rocprofv3 --pmc $grp --kernel-trace -d $R/gpurun_out/pmc_front -o pf --output-format csv -- python3 $R/bench.py --steps 2 --warmup 0 --inflight 1 --group 1 --cpu-sample 0 --no-e2e --no-verify > $R/gpurun_out/pmc_front.log 2>&1
  python3 - "$R/gpurun_out/pmc_front" "$PAT" >> $OUT <<'PY'
import csv, glob, collections, sys
f = glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True)[0]
pats = sys.argv[2].split()
acc = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.defaultdict(collections.Counter)
for r in csv.DictReader(open(f)):
    k = r['Kernel_Name'].split('(')[0].replace('void ', '').replace('scalce::', '')[:40]
    if not any(p in k for p in pats): continue
    acc[k][r['Counter_Name']] += float(r['Counter_Value']); calls[k][r['Counter_Name']] += 1
for k, v in sorted(acc.items()):
    print(k, {c: "%.4g" % (x / max(calls[k][c], 1)) for c, x in v.items()}, "per call")
PY
done
rm -rf $R/gpurun_out/pmc_front
cat $OUT
