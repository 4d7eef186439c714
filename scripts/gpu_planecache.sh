#!/bin/bash
# round 4: weight-image cache (one refresh launch per optimizer step): kernel test, model tests, step A/B with SMAAT_PLANE_CACHE
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/${1:-planecache}
mkdir -p "$OUT"
export SMAAT_REQUIRE_GPU=1
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py tests/test_torch_ops.py tests/test_autocast_yardstick.py -q -m gpu --tb=short -p no:cacheprovider -k "weight_planes or golden or bf16 or trajectory or traceable or ddp or step or yardstick or planes" > "$OUT/pytest.log" 2>&1
echo "pytest exit=$? $(grep -E 'passed|failed' "$OUT/pytest.log" | tail -1)"
grep -E "^(FAILED|ERROR)" "$OUT/pytest.log" | head
for pc in 1 0 1 0; do
  SMAAT_PLANE_CACHE=$pc timeout 300 python bench.py --steps 60 --warmup 5 --no-cpu-baseline --no-alt --no-latency --no-eager-baseline --no-side-configs --no-input-pipeline --no-power > "$OUT/bench_f32_pc$pc.json" 2> "$OUT/bench_f32_pc$pc.err"
  SMAAT_PLANE_CACHE=$pc timeout 300 python bench.py --precision bf16 --batch 64 --steps 40 --warmup 5 --no-cpu-baseline --no-alt --no-latency --no-eager-baseline --no-side-configs --no-input-pipeline --no-power > "$OUT/bench_bf16_pc$pc.json" 2> "$OUT/bench_bf16_pc$pc.err"
  echo "plane cache=$pc  $(python - <<PY
import json
def v(p):
    try:
        j = json.loads([l for l in open(p) if l.startswith("{")][-1]); return f"{j['value']} f/s {j['ms_per_step']} ms"
    except Exception as e: return f"n/a {e}"
print("f32", v("$OUT/bench_f32_pc$pc.json"), "| bf16", v("$OUT/bench_bf16_pc$pc.json"))
PY
)"
done
