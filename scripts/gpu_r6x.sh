#!/bin/bash
# Round 6: the final tree against itself with the round's two switches off (three-term fused forwards, torch's foreach Adam), interleaved
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/${1:-r6x}
mkdir -p "$OUT"
export SMAAT_REQUIRE_GPU=1
for rep in 1 2; do
for v in "1 one" "0 foreach" "0 one" "1 foreach"; do
  set -- $v
  SMAAT_FWD_ROWS_H=$1 SMAAT_ADAM=$2 timeout 600 python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-alt --no-latency --no-power --no-eager-baseline --no-side-configs --no-input-pipeline --no-profile 2>/dev/null | tail -1 > "$OUT/b.json"
  python - "$OUT/b.json" "$1" "$2" <<'PY' | tee -a "$OUT/bench_ab_round6_switches.txt"
import json,sys
d=json.loads(open(sys.argv[1]).read())
print("SMAAT_FWD_ROWS_H=%s SMAAT_ADAM=%-8s %.1f frames/s  %.3f ms/step" % (sys.argv[2], sys.argv[3], d["value"], d["ms_per_step"]))
PY
done
done
