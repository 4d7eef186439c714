#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/${1:-r6h}
mkdir -p "$OUT"
export SMAAT_REQUIRE_GPU=1
timeout 900 python -m pytest tests/test_gpu_f16_split.py -q -m gpu --tb=short -p no:cacheprovider -k "fused_backward" > "$OUT/pytest_fused_bwd.log" 2>&1
echo "pytest exit=$? $(tail -1 "$OUT/pytest_fused_bwd.log")"
B="python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-alt --no-latency --no-eager-baseline --no-input-pipeline --no-side-configs --no-power"
for i in 1 2; do
  for f in 1 0; do
    SMAAT_FUSED_BWD=$f timeout 600 $B > "$OUT/bench_fused${f}_$i.json" 2> "$OUT/bench_fused${f}_$i.err"
    python - <<PY
import json
try:
    j=json.loads([l for l in open("$OUT/bench_fused${f}_$i.json") if l.startswith("{")][-1]); print("bench fused=$f", j["value"], "frames/s", j["ms_per_step"], "ms")
    k=j.get("kernels") or {}
    for n,d in k.items():
        if "bwd_rows_h" in n: print("   ", n, d)
except Exception as e: print("bench parse error", e)
PY
  done
done
