#!/bin/bash
# Instruction counters of the coder kernels: rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS over a short default
# bench (three 50 M-read shards per grouped launch, the single job of the run through the four-blocks-per-wave kernel).
# Writes gpurun_out/<tag>_pmc_sq.json -- copy it to profiles/; bench.py reads roofline.kernel.issue from it.
#   bash tools/pmc_sq.sh r03_final
set -eu
TAG=${1:-r03_x}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/pmc_sq
timeout -k 10 500 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES --kernel-trace -d $R/gpurun_out/pmc_sq -o sq --output-format csv -- python3 $R/bench.py --steps 6 --warmup 0 --inflight 6 --group 3 --cpu-sample 0 --no-e2e --no-verify > $R/gpurun_out/pmc_sq.log 2>&1
cd $R && python3 - $TAG <<'PY'
import csv, glob, collections, json, sys
f = glob.glob('gpurun_out/pmc_sq/**/*counter_collection.csv', recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.Counter()
for r in csv.DictReader(open(f)):
    k = r['Kernel_Name'].split('(')[0].replace('void ', '')[:60]
    if 'ac_encode' not in k: continue
    acc[k][r['Counter_Name']] += float(r['Counter_Value'])
    if r['Counter_Name'] == 'SQ_WAVES': calls[k] += 1
SHARD = 50_000_000 * 100            # symbols of one shard
rows = []
for k, v in acc.items():
    # the timed steps go three shards per launch (eight blocks per wave, or one per lane); the one shard that runs alone first goes four per wave
    per_call = SHARD if ('16>' in k or k.endswith('ac_encode_k<false>')) else 3 * SHARD
    rows.append(dict(kernel=k, calls=calls[k], symbols=calls[k] * per_call, **{c: v[c] for c in ('SQ_INSTS_VALU', 'SQ_INSTS_SALU', 'SQ_INSTS_LDS', 'SQ_WAVES')}))
    print(k, calls[k], {c: round(v[c] / (calls[k] * per_call), 3) for c in ('SQ_INSTS_VALU', 'SQ_INSTS_SALU', 'SQ_INSTS_LDS')}, 'instructions per symbol')
json.dump(rows, open('gpurun_out/%s_pmc_sq.json' % sys.argv[1], 'w'), indent=1)
PY
rm -rf $R/gpurun_out/pmc_sq
