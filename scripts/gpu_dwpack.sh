#!/bin/bash
# round 4: plane packing of the row-streaming depthwise kernels (small planes share a wave): parity subset, then A/B timing
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/${1:-dwpack}
mkdir -p "$OUT"
export SMAAT_REQUIRE_GPU=1
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_bf16.py -q -m gpu --tb=short -p no:cacheprovider -k "dw3x3 or double_conv or bnred or dsconv_bwd" > "$OUT/pytest.log" 2>&1
echo "pytest exit=$? $(tail -1 "$OUT/pytest.log")"
grep -E "^(FAILED|ERROR)" "$OUT/pytest.log" | head
for pk in 1 0; do
  echo "== f32 batch 32 SMAAT_DW_PACK=$pk"
  SMAAT_DW_PACK=$pk timeout 300 python scripts/dw_bench.py 2>&1 | grep -v amdgpu.ids | tee "$OUT/dw_bench_f32_pack$pk.txt" | tail -4
  echo "== bf16 batch 64 SMAAT_DW_PACK=$pk"
  SMAAT_DW_PACK=$pk LB_ONLY=down4 timeout 300 python scripts/layer_bench_bf16.py 2>&1 | grep -v amdgpu.ids | tee "$OUT/layer_bench_bf16_down4_pack$pk.txt" | tail -4
done
timeout 600 python -m pytest tests/test_gpu_model.py -q -m gpu -x --tb=short -p no:cacheprovider -k "golden or bf16" > "$OUT/pytest_model.log" 2>&1
echo "pytest model exit=$? $(tail -1 "$OUT/pytest_model.log")"
for pk in 1 0; do
  SMAAT_DW_PACK=$pk timeout 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-alt --no-latency --no-eager-baseline --no-side-configs --no-input-pipeline --no-power > "$OUT/bench_pack$pk.json" 2> "$OUT/bench_pack$pk.err"
  echo "bench pack=$pk exit=$? $(python - <<PY
import json
try:
    d = json.loads(open("$OUT/bench_pack$pk.json").read().strip().splitlines()[-1]); print(d["value"], d["ms_per_step"])
except Exception as e: print("n/a", e)
PY
)"
done
echo done
