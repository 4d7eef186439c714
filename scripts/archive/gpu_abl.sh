#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/${1:-abl}
mkdir -p "$OUT"
for impl in 1 2; do
  SMAAT_PW_IMPL=$impl timeout 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "dsconv_fwd" --tb=short -p no:cacheprovider > "$OUT/k_dsconv_fwd_impl$impl.log" 2>&1
  echo "impl$impl dsconv_fwd exit=$? $(tail -1 "$OUT/k_dsconv_fwd_impl$impl.log")"
done
for abl in 0 1 2; do
  SMAAT_PW_IMPL=${ABL_IMPL:-2} SMAAT_PW_ABLATE=$abl timeout 300 python scripts/layer_bench.py > "$OUT/lb_abl$abl.txt" 2>&1
  echo "abl$abl exit=$?"; tail -1 "$OUT/lb_abl$abl.txt"
done
