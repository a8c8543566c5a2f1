#!/bin/bash
# WRITE_SIZE of emit_reads_k<true> with and without its q' half (timing/counter experiment; ablated outputs are wrong)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
for v in "" q p; do
  rm -rf /tmp/ew
  SCALCE_EMIT_ABLATE=$v timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/ew -o w -- python3 $R/bench.py --steps 2 --warmup 1 --group 1 --inflight 1 --cpu-sample 0 --no-e2e --no-verify > /tmp/ew.log 2>&1
  python3 - "$v" <<'P'
import csv,glob,sys
f=glob.glob('/tmp/ew/**/*counter_collection.csv',recursive=True)[0]
tot=0;n=0
for r in csv.DictReader(open(f)):
    if 'emit_reads_k' in r['Kernel_Name'] and r['Counter_Name']=='WRITE_SIZE': tot+=float(r['Counter_Value']); n+=1
print("ablate '%s': emit_reads_k WRITE_SIZE %.2f GB per call (%d calls)" % (sys.argv[1], tot*1024/n/1e9, n))
P
done
