#!/bin/bash
# parity (dsconv/pointwise groups) + layer bench for a list of SMAAT_PW_IMPL values
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/${1:-impl}
mkdir -p "$OUT"
for impl in ${IMPLS:-4}; do
  SMAAT_PW_IMPL=$impl timeout 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "dsconv_fwd or pointwise_fwd" --tb=short -p no:cacheprovider > "$OUT/k_impl$impl.log" 2>&1
  echo "impl$impl tests exit=$? $(tail -1 "$OUT/k_impl$impl.log")"
  for abl in ${ABLS:-0}; do
    SMAAT_PW_IMPL=$impl SMAAT_PW_ABLATE=$abl timeout 300 python scripts/layer_bench.py > "$OUT/lb_impl${impl}_abl$abl.txt" 2>&1
    echo "impl$impl abl$abl exit=$? $(tail -1 "$OUT/lb_impl${impl}_abl$abl.txt")"
  done
done
