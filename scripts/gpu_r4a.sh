#!/bin/bash
# round 4, first call: the whole GPU suite (incl. the new batch-32 / VOC batch-16 reference fixtures and the
# reference-autocast yardstick), smoke, the default bench line, and the self-launching multi-rank entry of bench.py
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/${1:-r4a}
mkdir -p "$OUT"
export SMAAT_REQUIRE_GPU=1
timeout 900 python -m pytest tests -q -m gpu --tb=short -p no:cacheprovider --durations=15 > "$OUT/pytest_gpu.log" 2>&1
echo "pytest exit=$? $(tail -1 "$OUT/pytest_gpu.log")"
grep -E "^(FAILED|ERROR)" "$OUT/pytest_gpu.log" | head -40
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > "$OUT/smoke.log" 2>&1
echo "smoke exit=$? $(tail -1 "$OUT/smoke.log")"
# bench.py --gpus 2 without a launcher: (a) over RCCL it must spawn and then say that 2 devices are needed;
# (b) with the gloo testing backend both ranks share the GPU and the whole multi-rank control flow runs
timeout 300 python bench.py --gpus 2 --steps 2 --warmup 1 > "$OUT/bench_gpus2_nccl.log" 2>&1
echo "bench --gpus 2 (nccl, 1 device) exit=$? $(grep -m1 'needs 2 devices' "$OUT/bench_gpus2_nccl.log")"
SMAAT_BENCH_BACKEND=gloo timeout 600 python bench.py --gpus 2 --steps 3 --warmup 1 --no-profile --no-latency --no-power \
    > "$OUT/bench_gpus2_gloo.log" 2>&1
echo "bench --gpus 2 (gloo, shared device) exit=$? $(grep -o '"value": [0-9.]*, "unit": "frames/s", "n_gpus": 2, "rccl_ranks": 2' "$OUT/bench_gpus2_gloo.log" | head -1)"
timeout 900 python bench.py > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"
echo "bench exit=$? $(python - <<PY
import json
try:
    j = json.loads([l for l in open("$OUT/bench_default.json") if l.startswith("{")][-1])
    print(j["value"], j["unit"], j["ms_per_step"], "ms; bf16", j["configs"]["bf16_b64"].get("value"), "voc", j["configs"]["voc_b16"].get("value"),
          "lat", j["fwd_latency"].get("hipgraph_ms"), "fed", j["input_pipeline_fed"].get("value"), "fed_h5", j["input_pipeline_fed_hdf5"])
except Exception as e:
    print("parse error", e)
PY
)"
ls gpurun_out/*.json 2>/dev/null | head
