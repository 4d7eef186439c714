#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/${1:-dwb}
mkdir -p "$OUT"
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "dw3x3_bwd" --tb=short -p no:cacheprovider > "$OUT/k_dwb.log" 2>&1
echo "dwb tests exit=$? $(tail -1 "$OUT/k_dwb.log")"
for v in 0 1; do
  SMAAT_DWB_STRIP=$v timeout 300 python scripts/layer_bench.py > "$OUT/lb_strip$v.txt" 2>&1
  echo "strip$v exit=$? $(tail -1 "$OUT/lb_strip$v.txt")"
done
