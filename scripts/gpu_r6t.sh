#!/bin/bash
# Round 6: one-launch Adam -- GPU tests, interleaved step A/B against torch's foreach and fused implementations
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/${1:-r6t}
mkdir -p "$OUT"
export SMAAT_REQUIRE_GPU=1
timeout 600 python -m pytest tests/test_optim.py -x -q -m gpu > "$OUT/pytest_optim.log" 2>&1
echo "pytest optim exit=$? $(tail -1 "$OUT/pytest_optim.log")"
for v in one foreach fused one foreach fused; do
  SMAAT_ADAM=$v timeout 600 python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-alt --no-latency --no-power --no-eager-baseline --no-side-configs --no-input-pipeline --no-profile 2>/dev/null | tail -1 > "$OUT/bench_$v.json"
  python - "$OUT/bench_$v.json" "$v" <<'PY' | tee -a "$OUT/bench_ab_adam.txt"
import json,sys
d=json.loads(open(sys.argv[1]).read())
print("SMAAT_ADAM=%-8s %.1f frames/s  %.3f ms/step  final loss %s" % (sys.argv[2], d["value"], d["ms_per_step"], d["config"]["final_loss"]))
PY
done
