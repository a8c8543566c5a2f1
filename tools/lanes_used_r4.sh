cd "$(dirname "$0")/.."
for lu in 40 48 56 64; do
  out=$(SCALCE_AC_LANES_USED=$lu python bench.py --steps 20 --warmup 2 --no-e2e --no-verify --cpu-sample 0 2>/dev/null)
  python - "$out" $lu <<'P'
import json,sys
j=json.loads(sys.argv[1]); print("lanes used %s: %.1f ms per shard, launch %.0f ms" % (sys.argv[2], j["ms_per_step"], j["roofline"]["kernel"]["launch_ms"]))
P
done
