#!/bin/bash
# gpu_final.sh without the test suite: default bench line, rocprofv3 statistics + counter passes (f32, bf16), traffic record,
# multi-rank pre-flight of bench.py on one device (gloo, testing backend)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
TAG=${1:-final2}
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
export SMAAT_REQUIRE_GPU=1
timeout 900 python bench.py > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"
echo "bench exit=$? $(python - <<PY
import json
try:
    j = json.loads([l for l in open("$OUT/bench_default.json") if l.startswith("{")][-1])
    print(j["value"], j["unit"], j["ms_per_step"], "ms; bf16", j["configs"]["bf16_b64"].get("value"), "voc", j["configs"]["voc_b16"].get("value"),
          "lat", j["fwd_latency"].get("hipgraph_ms"), "power", j.get("power"))
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
SMAAT_BENCH_BACKEND=gloo timeout 600 python bench.py --gpus 2 --steps 3 --warmup 1 --no-profile --no-latency --no-power --no-cpu-baseline --no-alt --no-eager-baseline --no-input-pipeline --no-side-configs > "$OUT/bench_gpus2_gloo_shared_device.log" 2>&1
echo "2-rank gloo preflight exit=$? $(grep -o '"value": [0-9.]*' "$OUT/bench_gpus2_gloo_shared_device.log" | head -1)"
timeout 300 python bench.py --gpus 2 --steps 3 --warmup 1 --no-profile --no-latency --no-power --no-cpu-baseline --no-alt --no-eager-baseline --no-input-pipeline --no-side-configs > "$OUT/bench_gpus2_nccl_one_device.log" 2>&1
echo "2-rank nccl on one device exit=$? $(tail -2 "$OUT/bench_gpus2_nccl_one_device.log" | cut -c1-200)"
du -sh gpurun_out
