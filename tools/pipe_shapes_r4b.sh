#!/bin/bash
# bench.py at the driver's 20 steps for (shards per coder launch):(coder streams) shapes, twelve shards in flight
cd "$(dirname "$0")/.."
for v in ${1:-6:2 4:3 3:4 2:6}; do
  g=${v%%:*}; st=${v##*:}
  out=$(SCALCE_BENCH_CODER_STREAMS=$st python bench.py --steps 20 --warmup 2 --group $g --inflight 12 --no-e2e --no-verify --cpu-sample 0 2>/dev/null)
  python - "$out" $v <<'P'
import json,sys
j=json.loads(sys.argv[1]); print("group:streams %s: %.1f ms per shard, launch %.0f ms, in flight %s, per launch %s" % (sys.argv[2], j["ms_per_step"], j["roofline"]["kernel"]["launch_ms"], j["config"].get("shards_in_flight"), j["config"].get("shards_per_coder_launch")), flush=True)
P
done
