#!/bin/bash
# checkpoint: whole GPU suite + default bench line (no profiling passes).  usage: gpu_check.sh <tag>
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/${1:-check}
mkdir -p "$OUT"
export SMAAT_REQUIRE_GPU=1
timeout 1200 python -m pytest tests -q -m gpu --tb=short -p no:cacheprovider > "$OUT/pytest_gpu.log" 2>&1
echo "pytest exit=$? $(tail -1 "$OUT/pytest_gpu.log")"
grep -E "^(FAILED|ERROR)|Memory access" "$OUT/pytest_gpu.log" | head -20
timeout 900 python bench.py --no-cpu-baseline --no-eager-baseline > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"
echo "bench exit=$? $(python - <<PY
import json
try:
    j = json.loads([l for l in open("$OUT/bench_default.json") if l.startswith("{")][-1])
    print(j["value"], j["unit"], j["ms_per_step"], "ms; bf16", j["configs"]["bf16_b64"].get("value"), "voc", j["configs"]["voc_b16"].get("value"),
          "lat", j["fwd_latency"].get("hipgraph_ms"), "peak GiB", j.get("hbm_peak_allocated_gib"))
except Exception as e:
    print("parse error", e)
PY
)"
