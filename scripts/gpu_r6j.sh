#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/${1:-r6j}
mkdir -p "$OUT"
export SMAAT_REQUIRE_GPU=1
timeout 1500 python -m pytest tests -q -m gpu --tb=short -p no:cacheprovider --durations=8 > "$OUT/pytest_gpu.log" 2>&1
echo "pytest exit=$? $(grep -E "passed|failed" "$OUT/pytest_gpu.log" | tail -1)"
grep -E "^(FAILED|ERROR)" "$OUT/pytest_gpu.log" | head
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > "$OUT/smoke.log" 2>&1
echo "smoke exit=$? $(tail -1 "$OUT/smoke.log")"
timeout 900 python bench.py > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"
python - <<PY
import json
j = json.loads([l for l in open("$OUT/bench_default.json") if l.startswith("{")][-1])
print("bench", j["value"], j["ms_per_step"], "traffic", j["roofline"].get("traffic"), "frac", j["roofline"]["frac"], "power samples", (j.get("power") or {}).get("samples"))
PY
