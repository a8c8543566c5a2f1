#!/bin/bash
# usage: tools/pipe_shapes_r4.sh  -> one line per pipeline shape: shards per launch / coder streams / steps -> ms per shard
# (round 4: every shard in flight owns its text, so ~11 fit; a lanes launch takes ~0.6 s whatever it holds)
cd "$(dirname "$0")/.."
run() { # group streams steps blocks_per_wg queues [inflight]
  local out
  out=$(GPU_MAX_HW_QUEUES=$5 SCALCE_AC_BLOCKS_PER_WG=$4 SCALCE_BENCH_CODER_STREAMS=$2 python bench.py --group $1 --inflight ${6:-11} --steps $3 --warmup 2 --no-e2e --no-verify --cpu-sample 0 2>/dev/null)
  python - "$out" "$@" <<'P'
import json,sys
j=json.loads(sys.argv[1]); print("group %s streams %s steps %s bpw %s queues %s: %.1f ms per shard, in flight %d, launch %.0f ms, hbm %.0f GB" % (*sys.argv[2:7], j["ms_per_step"], j["config"]["shards_in_flight"], j["roofline"]["kernel"]["launch_ms"], j["config"]["hbm_used_gb"]))
P
}
IFS=";" read -ra CF <<< "${SHAPES:-3 3 20 0 8;2 4 20 64 8;2 5 20 64 12;1 8 20 64 12;1 10 20 64 16;3 3 44 0 8;2 5 44 64 12;1 10 44 64 16}"
for cfg in "${CF[@]}"; do
  run $cfg
done
