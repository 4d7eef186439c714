#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/${1:-split}
mkdir -p "$OUT"
for m in ${MODES:-0 3 2}; do
  SMAAT_SPLIT=$m timeout 300 python scripts/split_accuracy.py > "$OUT/acc_$m.txt" 2>&1; echo "acc $m exit=$?"; grep split= "$OUT/acc_$m.txt"
done
for m in ${TESTMODES:-3}; do
  SMAAT_SPLIT=$m timeout 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "${TESTK:-pointwise_wgrad}" --tb=short -p no:cacheprovider > "$OUT/k_split$m.log" 2>&1
  echo "split$m tests exit=$? $(tail -1 "$OUT/k_split$m.log")"
done
for m in ${MODES:-0 3 2}; do
  SMAAT_SPLIT=$m timeout 300 python scripts/layer_bench.py > "$OUT/lb_split$m.txt" 2>&1
  echo "lb split$m exit=$? $(tail -1 "$OUT/lb_split$m.txt")"
done
