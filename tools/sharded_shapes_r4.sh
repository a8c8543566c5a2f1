cd "$(dirname "$0")/.."
run() { # group inflight streams bpw steps
  out=$(SCALCE_BENCH_FORCE_SHARDED=1 SCALCE_AC_BLOCKS_PER_WG=$4 SCALCE_BENCH_CODER_STREAMS=$3 python bench.py --group $1 --inflight $2 --steps $5 --warmup 2 --no-e2e --no-verify --cpu-sample 0 2>/dev/null)
  python - "$out" "$@" <<'P'
import json,sys
j=json.loads(sys.argv[1]); print("sharded: group %s inflight %s streams %s bpw %s steps %s: %.1f ms per shard, launch %.0f ms, hbm %.0f GB" % (*sys.argv[2:7], j["ms_per_step"], j["roofline"]["kernel"]["launch_ms"], j["config"]["hbm_used_gb"]))
P
}
run 3 6 1 8 12
run 3 8 1 8 12
run 2 8 3 64 12
run 2 8 4 64 12
run 3 8 2 64 12
