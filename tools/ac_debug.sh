#!/bin/bash
# profiling only: time ac_encode_k with parts of the workgroup disabled (output is garbage in those runs).
# 1 = chain idles (writes empty outcomes), 2 = no pack, 6 = no pack and no gather.  Never run 4 without 2.
for d in 0 1 2 6; do
  echo "debug=$d" >> gpurun_out/ac_debug.log
  SCALCE_AC_DEBUG=$d timeout -k 5 60 python bench.py --reads 20000000 --steps 1 --warmup 1 --cpu-sample 0 2>/dev/null | python -c "
import sys,json
j=json.loads(sys.stdin.readline()); print('  ac launch ms', j['roofline']['launch_ms'], 'step ms', j['ms_per_step'])" >> gpurun_out/ac_debug.log 2>&1
done
