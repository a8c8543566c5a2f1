#!/bin/bash
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r5_full; mkdir -p $O
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > $O/tests.txt 2>&1; echo "tests rc $?" >> $O/tests.txt
tail -4 $O/tests.txt
timeout -k 10 600 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "bench rc $?"
python - <<'P'
import json
j=json.loads(open('gpurun_out/r5_full/bench.json').read().strip().splitlines()[-1])
print({k:j[k] for k in ('value','ms_per_step','parity_checked','value_single_job')}, j['config']['shards_in_flight'], j['decode'] and j['decode']['value'], j['e2e'] and {k:v.get('value') for k,v in j['e2e'].items() if isinstance(v,dict)}, j['table_scale'] and j['table_scale']['ns_per_read'], j['cpu_baseline'] and (j['cpu_baseline']['value'], j['cpu_baseline'].get('configs0_gz')))
P
