#!/bin/bash
# fused depthwise->split-GEMM kernel: correctness subset, per-layer table, eval latency
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/${1:-fused}
mkdir -p "$OUT"
export SMAAT_REQUIRE_GPU=1
timeout 300 python -m pytest tests/test_gpu_kernels.py -q -m gpu --tb=short -p no:cacheprovider -k "dsconv or bn_partial" > "$OUT/pytest.log" 2>&1
echo "pytest exit=$? $(tail -1 "$OUT/pytest.log")"
grep -E "^(FAILED|ERROR)" "$OUT/pytest.log" | head
timeout 300 python scripts/layer_bench.py > "$OUT/layer_bench.txt" 2>&1
echo "layer exit=$?"; grep -E "inc|down1|up3|up4|total|name" "$OUT/layer_bench.txt" | head -20
timeout 300 python scripts/eval_latency.py > "$OUT/eval_latency.txt" 2>&1
echo "eval exit=$?"; tail -8 "$OUT/eval_latency.txt"
SMAAT_DSS_ROWMAP=1 timeout 300 python scripts/layer_bench.py > "$OUT/layer_bench_rowmap.txt" 2>&1
echo "layer(rowmap) exit=$?"; grep -E "inc|down1|up3|up4" "$OUT/layer_bench_rowmap.txt" | sed 's/.*FUSED/FUSED/' | head -20
