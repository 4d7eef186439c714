#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/${1:-cbam}
mkdir -p "$OUT"
timeout 200 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "cbam or pool_upsample or upsample" --tb=short -p no:cacheprovider > "$OUT/k_cbam.log" 2>&1
echo "cbam kernels exit=$? $(tail -1 "$OUT/k_cbam.log")"
timeout 300 python -m pytest tests/test_gpu_model.py -q -m gpu -x --tb=short -p no:cacheprovider > "$OUT/model.log" 2>&1
echo "model exit=$? $(tail -1 "$OUT/model.log")"
timeout 300 python bench.py --steps 10 --warmup 3 --no-alt --no-latency --no-cpu-baseline > "$OUT/bench.json" 2> "$OUT/bench.err"
python - "$OUT" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1] + "/bench.json").read().strip().splitlines()[-1])
k = d["kernels"]
print("bench", d["value"], d["ms_per_step"], "ups_bwd", k.get("smaat_upsample2x_bwd"), "ups_fwd", k.get("smaat_upsample2x_fwd"))
PY
