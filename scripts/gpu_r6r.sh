#!/bin/bash
# Round 6: last validation of the committed tree -- GPU suite, smoke, default bench line (traffic record of this build)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/${1:-r6r}
mkdir -p "$OUT"
export SMAAT_REQUIRE_GPU=1
timeout 2400 python -m pytest tests -x -q -m gpu > "$OUT/pytest_gpu.log" 2>&1
echo "pytest gpu exit=$? $(grep -E 'passed|failed' "$OUT/pytest_gpu.log" | tail -1)"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > "$OUT/smoke.log" 2>&1
echo "smoke exit=$? $(tail -1 "$OUT/smoke.log")"
timeout 900 python bench.py > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"
echo "bench exit=$?"
python - "$OUT/bench_default.json" <<'PY'
import json,sys
j=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
r=j["roofline"]
print(j["value"], j["unit"], j["ms_per_step"], "ms | roofline frac", r["frac"], "traffic", r["traffic"], "| bf16", j["configs"]["bf16_b64"].get("value"), "voc", j["configs"]["voc_b16"].get("value"), "lat", j["fwd_latency"].get("hipgraph_ms"))
for c in r.get("other_classes", []):
    if "smaat_dsconv_fwd_rows_h" in c["entry_points"]:
        print("rows_h class:", c["ms_per_step"], "ms", c["algorithmic_tflops"], "TFLOP/s alg", c["algorithmic_gbs"], "GB/s", "traffic", c["traffic"])
PY
