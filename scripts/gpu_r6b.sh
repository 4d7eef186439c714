#!/bin/bash
# Round 6 (b): ISA-level root-cause experiments + the whole GPU suite + a default bench on the padded-walk kernels
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/${1:-r6b}
mkdir -p "$OUT"
export SMAAT_REQUIRE_GPU=1
RC_REPS=32 timeout 900 python scripts/probes/r6_dswgrad_rootcause_run.py > "$OUT/dswgrad_rootcause_isa.txt" 2>&1
echo "rootcause exit=$?"; cat "$OUT/dswgrad_rootcause_isa.txt"
timeout 1500 python -m pytest tests -q -m gpu --tb=short -p no:cacheprovider > "$OUT/pytest_gpu.log" 2>&1
echo "pytest exit=$? $(tail -1 "$OUT/pytest_gpu.log")"
grep -E "^(FAILED|ERROR)" "$OUT/pytest_gpu.log" | head -20
for i in 1 2; do
timeout 600 python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-alt --no-latency --no-eager-baseline --no-input-pipeline --no-side-configs > "$OUT/bench_quick_$i.json" 2> "$OUT/bench_quick_$i.err"
python - <<PY
import json
try:
    j=json.loads([l for l in open("$OUT/bench_quick_$i.json") if l.startswith("{")][-1]); print("bench", j["value"], "frames/s", j["ms_per_step"], "ms")
except Exception as e: print("bench parse error", e)
PY
done
