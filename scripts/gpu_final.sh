#!/bin/bash
# The round's validation on one box: whole GPU suite (what the driver runs) + smoke + default bench line, then rocprofv3
# kernel statistics + FETCH / WRITE / MFMA counter passes of the SAME build in f32 and in mixed precision, and the traffic
# record bench.py quotes.  usage: gpu_final.sh <tag>     (everything lands under gpurun_out/<tag>/)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
TAG=${1:-final}
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
export SMAAT_REQUIRE_GPU=1
timeout 1200 python -m pytest tests -q -m gpu --tb=short -p no:cacheprovider --durations=10 > "$OUT/pytest_gpu.log" 2>&1
echo "pytest exit=$? $(tail -1 "$OUT/pytest_gpu.log")"
grep -E "^(FAILED|ERROR)|Memory access" "$OUT/pytest_gpu.log" | head -20
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > "$OUT/smoke.log" 2>&1
echo "smoke exit=$? $(tail -1 "$OUT/smoke.log")"
timeout 900 python bench.py > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"
echo "bench exit=$? $(python - <<PY
import json
try:
    j = json.loads([l for l in open("$OUT/bench_default.json") if l.startswith("{")][-1])
    print(j["value"], j["unit"], j["ms_per_step"], "ms; bf16", j["configs"]["bf16_b64"].get("value"), "voc", j["configs"]["voc_b16"].get("value"),
          "lat", j["fwd_latency"].get("hipgraph_ms"), "fed", j["input_pipeline_fed"].get("value"), "fed_h5", j["input_pipeline_fed_hdf5"].get("value"),
          "peak GiB", j["hbm_peak_allocated_gib"])
except Exception as e:
    print("parse error", e)
PY
)"
bash scripts/prof_round.sh $TAG/prof_f32 > "$OUT/prof_f32.log" 2>&1
echo "prof f32 done: $(grep -c exit= "$OUT/prof_f32.log") passes"
bash scripts/prof_round.sh $TAG/prof_bf16 "--precision bf16 --batch 64" > "$OUT/prof_bf16.log" 2>&1
echo "prof bf16 done: $(grep -c exit= "$OUT/prof_bf16.log") passes"
python scripts/make_traffic_json.py "$OUT/prof_f32/summary.txt" "$OUT/prof_bf16/summary.txt" > "$OUT/traffic.log" 2>&1
cp profiles/hbm_traffic.json "$OUT/hbm_traffic.json"
tail -2 "$OUT/traffic.log"
