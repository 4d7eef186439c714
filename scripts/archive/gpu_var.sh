#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/${1:-var}
mkdir -p "$OUT"
SMAAT_PW_IMPL=3 timeout 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "dsconv_fwd or pointwise_fwd" --tb=short -p no:cacheprovider > "$OUT/k_impl3.log" 2>&1
echo "impl3 tests exit=$? $(tail -1 "$OUT/k_impl3.log")"
for v in hip varB varC varD; do
  for impl in 2 3; do
    SMAAT_LIB=$PWD/smaat_unet_amd/libsmaat_$v.so SMAAT_PW_IMPL=$impl LB_SKIP_BWD=1 timeout 300 python scripts/layer_bench.py > "$OUT/lb_${v}_impl$impl.txt" 2>&1
    echo "$v impl$impl exit=$? $(tail -1 "$OUT/lb_${v}_impl$impl.txt")"
  done
done
