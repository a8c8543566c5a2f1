set -eu
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/pmc_sq
timeout -k 10 500 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES --kernel-trace -d $R/gpurun_out/pmc_sq -o sq --output-format csv -- python3 $R/bench.py --steps 3 --warmup 0 --inflight 6 --group 3 --cpu-sample 0 --no-e2e > $R/gpurun_out/pmc_sq.log 2>&1
cd $R && python3 - <<'PY'
import csv,glob,collections
f=glob.glob('gpurun_out/pmc_sq/**/*counter_collection.csv',recursive=True)[0]
acc=collections.defaultdict(lambda: collections.defaultdict(float)); calls=collections.Counter()
for r in csv.DictReader(open(f)):
    k=r['Kernel_Name'].split('(')[0][:50]
    if 'ac_encode' not in k: continue
    acc[k][r['Counter_Name']]+=float(r['Counter_Value'])
    if r['Counter_Name']=='SQ_WAVES': calls[k]+=1
for k,v in acc.items(): print(k, calls[k], dict(v))
PY
