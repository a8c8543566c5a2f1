#!/bin/bash
# Counter evidence for the decode leg ("the decoder's recurrence is an ISA reading, not a counter"): kernel time of
# ac_decode_tight_k on one 50 M x 100 bp shard, then its instruction and wave-cycle counters in separate --pmc passes.
#   bash tools/decode_profile.sh r05   -> gpurun_out/r05_decode_kernel_stats.csv, gpurun_out/r05_decode_pmc.json
set -eu
TAG=${1:-r05}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/prof_dec; rm -rf $O; mkdir -p $O
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o st -- python3 $R/tools/decode_time.py 50000000 1 > $O/stats.log 2>&1
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD --kernel-trace --output-format csv -d $O/pmc1 -o p -- python3 $R/tools/decode_time.py 50000000 1 > $O/pmc1.log 2>&1
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $O/pmc2 -o p -- python3 $R/tools/decode_time.py 50000000 1 > $O/pmc2.log 2>&1
cd $R
python3 tools/prof_summary.py stats $O/stats gpurun_out/${TAG}_decode_kernel_stats.csv | tail -1
python3 - $TAG <<'PY'
import csv, glob, json, sys, collections
acc = collections.defaultdict(float)
for d in ("pmc1", "pmc2"):
    for f in glob.glob(f"gpurun_out/prof_dec/{d}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "ac_decode_tight_k" in r["Kernel_Name"]:
                acc[r["Counter_Name"]] += float(r["Counter_Value"])
sym = 50_000_000 * 100
out = {"kernel": "ac_decode_tight_k", "symbols": sym, "blocks": 477, **acc}
for k in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD"):
    if k in acc: out[k + "_per_symbol"] = round(acc[k] / sym, 3)
json.dump(out, open("gpurun_out/%s_decode_pmc.json" % sys.argv[1], "w"), indent=1)
print(json.dumps({k: v for k, v in out.items() if k.endswith("per_symbol")}), {k: acc[k] for k in acc if "CYCLES" in k or "WAIT" in k})
PY
grep "ns per symbol" $O/stats.log | tail -1
rm -rf $O
