#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/${1:-cfg}
mkdir -p "$OUT"
for c in ${CFGS:-0 1 8}; do
  SMAAT_PWS_CFG=$c timeout 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "pointwise_fwd_split or pointwise_wgrad" --tb=short -p no:cacheprovider > "$OUT/k_cfg$c.log" 2>&1
  echo "cfg$c tests exit=$? $(tail -1 "$OUT/k_cfg$c.log")"
  SMAAT_PWS_CFG=$c LB_ONLY="${LB_ONLY:-}" timeout 300 python scripts/layer_bench.py > "$OUT/lb_cfg$c.txt" 2>&1
  echo "cfg$c exit=$? $(tail -1 "$OUT/lb_cfg$c.txt")"
done
