#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/${1:-xcd}
mkdir -p "$OUT"
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "dsconv_fwd or pointwise_fwd" --tb=short -p no:cacheprovider > "$OUT/k.log" 2>&1
echo "tests exit=$? $(tail -1 "$OUT/k.log")"
for abl in 0 16; do
  SMAAT_SPLIT=0 SMAAT_PW_ABLATE=$abl timeout 300 python scripts/layer_bench.py > "$OUT/lb_f32_abl$abl.txt" 2>&1
  echo "f32 abl$abl exit=$? $(tail -1 "$OUT/lb_f32_abl$abl.txt")"
done
timeout 300 python scripts/layer_bench.py > "$OUT/lb_split.txt" 2>&1
echo "split exit=$? $(tail -1 "$OUT/lb_split.txt")"
