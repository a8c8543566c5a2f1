#!/bin/bash
# bench.py at the driver's 20 steps for (sets of four waves per coder workgroup):(lanes in use per set) variants
cd "$(dirname "$0")/.."
for v in ${1:-1:48 2:48 2:40}; do
  s=${v%%:*}; lu=${v##*:}
  out=$(SCALCE_AC_SETS=$s SCALCE_AC_LANES_USED=$lu python bench.py --steps 20 --warmup 2 --no-e2e --no-verify --cpu-sample 0 2>/dev/null)
  python - "$out" $v <<'P'
import json,sys
j=json.loads(sys.argv[1]); print("sets:lanes %s: %.1f ms per shard, launch %.0f ms, in flight %s, per launch %s" % (sys.argv[2], j["ms_per_step"], j["roofline"]["kernel"]["launch_ms"], j["config"].get("shards_in_flight"), j["config"].get("shards_per_coder_launch")), flush=True)
P
done
