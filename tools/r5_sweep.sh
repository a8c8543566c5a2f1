#!/bin/bash
# round 5: pipeline shapes with coding in place (slots are cheap now): shards per launch / in flight, two sets of waves per CU,
# blocks per workgroup.  usage (GPU box): bash tools/r5_sweep.sh
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r5_sweep; mkdir -p $O
: > $O/summary.txt
run() { # label args... (env through "env")
  local label="$1"; shift
  "$@" > $O/last.txt 2>&1
  python - "$label" >> $O/summary.txt <<'P'
import sys,json
lab=sys.argv[1]
try:
    j=[json.loads(l) for l in open('gpurun_out/r5_sweep/last.txt') if l.startswith('{')][-1]
    c=j['config']
    print('%-44s %7.2f ms  slots %2d G %d streams %d hbm %3.0f launch %4.0f ms' % (lab, j['ms_per_step'], c['shards_in_flight'], c['shards_per_coder_launch'], c['coder_streams'], c['hbm_used_gb'], j['roofline']['kernel']['launch_ms']))
except Exception as e:
    print(lab, 'FAILED', repr(e)[:100], open('gpurun_out/r5_sweep/last.txt').read()[-300:].replace('\n',' | '))
P
  tail -1 $O/summary.txt
}
B="python bench.py --steps 20 --warmup 5 --no-e2e --cpu-sample 0 --no-verify"
run "15 / 6 lanes 32"                 env SCALCE_AC_LANES_USED=32 $B --group 6 --inflight 15
run "15 / 6 lanes 24"                 env SCALCE_AC_LANES_USED=24 $B --group 6 --inflight 15
run "15 / 6 lanes 28"                 env SCALCE_AC_LANES_USED=28 $B --group 6 --inflight 15
run "15 / 6 lanes 36"                 env SCALCE_AC_LANES_USED=36 $B --group 6 --inflight 15
run "16 / 6 lanes 32"                 env SCALCE_AC_LANES_USED=32 $B --group 6 --inflight 16
run "18 / 6 lanes 32 2 streams"       env SCALCE_AC_LANES_USED=32 SCALCE_BENCH_CODER_STREAMS=2 $B --group 6 --inflight 18
run "18 / 6 lanes 32 3 streams"       env SCALCE_AC_LANES_USED=32 $B --group 6 --inflight 18
run "16 / 7 lanes 32"                 env SCALCE_AC_LANES_USED=32 $B --group 7 --inflight 16
run "16 / 8 lanes 32"                 env SCALCE_AC_LANES_USED=32 $B --group 8 --inflight 16
run "15 / 6 lanes 32 no side"         env SCALCE_AC_LANES_USED=32 SCALCE_BENCH_NO_SIDE=1 $B --group 6 --inflight 15
run "15 / 6 lanes 32, 36 steps"       env SCALCE_AC_LANES_USED=32 python bench.py --steps 36 --warmup 5 --no-e2e --cpu-sample 0 --no-verify --group 6 --inflight 15
cat $O/summary.txt
