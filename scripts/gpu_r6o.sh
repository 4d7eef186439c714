#!/bin/bash
# Round 6: the fused row-walking forward on the two-term fp16 split, wired into the step -- its tests, the whole network tests,
# and the step time with / without it
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/${1:-r6o}
mkdir -p "$OUT"
export SMAAT_REQUIRE_GPU=1
timeout 1500 python -m pytest tests/test_gpu_f16_split.py -x -q -m gpu -k "rows_forward_h or cbam_apply_amax or upsample2x_fwd_amax" > "$OUT/pytest_rows_h.log" 2>&1
echo "pytest rows_h exit=$? $(tail -1 "$OUT/pytest_rows_h.log")"
timeout 1500 python -m pytest tests/test_gpu_model.py tests/test_eval_and_big.py tests/test_strict_blocks.py -x -q -m gpu > "$OUT/pytest_model.log" 2>&1
echo "pytest model exit=$? $(tail -1 "$OUT/pytest_model.log")"
for v in 1 0 1 0; do
  SMAAT_FWD_ROWS_H=$v timeout 600 python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-alt --no-latency --no-power --no-eager-baseline --no-side-configs --no-input-pipeline 2>/dev/null | tail -1 > "$OUT/bench_h$v.json"
  python - "$OUT/bench_h$v.json" "$v" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read())
print("SMAAT_FWD_ROWS_H=%s  %.1f frames/s  %.3f ms/step" % (sys.argv[2], d["value"], d["ms_per_step"]))
PY
done
