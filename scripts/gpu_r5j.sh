#!/bin/bash
# Round 5: whole GPU suite + smoke on the three-pass / channel-split attention build, then interleaved A/B of the step
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
TAG=${1:-r5j}
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
export SMAAT_REQUIRE_GPU=1
timeout 1500 python -m pytest tests -q -m gpu --tb=short -p no:cacheprovider --durations=8 > "$OUT/pytest_gpu.log" 2>&1
echo "suite exit=$? $(tail -1 "$OUT/pytest_gpu.log")"
grep -E "^(FAILED|ERROR)|Memory access" "$OUT/pytest_gpu.log" | head -30
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > "$OUT/smoke.log" 2>&1
echo "smoke exit=$? $(tail -1 "$OUT/smoke.log")"
B="python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-profile --no-alt --no-latency --no-eager-baseline --no-input-pipeline --no-side-configs"
for rep in 1 2; do
  for f in 1 0; do
    SMAAT_CBAM_THREE_PASS=$f timeout 300 $B > "$OUT/bench_three_${f}_$rep.json" 2> "$OUT/bench_three_${f}_$rep.err"
    echo "THREE_PASS=$f rep $rep: $(python -c "
import json
try:
    j=json.loads([l for l in open('$OUT/bench_three_${f}_$rep.json') if l.startswith('{')][-1]); print(j['value'], j['ms_per_step'])
except Exception as e: print('parse error', e)
")"
  done
done
ls gpurun_out/*tie_flips.json 2>/dev/null
