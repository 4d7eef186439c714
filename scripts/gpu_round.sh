#!/bin/bash
# One GPU-box session: per-kernel parity tests (each group in its own process so that a
# fault in one kernel does not hide the others), module/UNet parity, smoke, short bench.
# Everything is logged under gpurun_out/.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/${1:-r1}
mkdir -p "$OUT"
rocminfo 2>/dev/null | grep -E "Marketing Name|gfx" | head -4 > "$OUT/device.txt"
nproc >> "$OUT/device.txt"
for grp in dsconv_fwd "pointwise_fwd and not split" pointwise_fwd_split dw3x3_fwd dsconv_wgrad pointwise_wgrad "dw3x3_bwd and not bnred" dw3x3_bwd_bnred "bn_chain or bn_eval" misc pool_upsample cbam; do
  timeout 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "$grp" --tb=short -p no:cacheprovider \
      > "$OUT/k_${grp// /_}.log" 2>&1
  echo "$grp exit=$? $(tail -1 "$OUT/k_${grp// /_}.log")"
done
timeout 900 python -m pytest tests/test_gpu_model.py -q -m gpu --tb=short -p no:cacheprovider > "$OUT/model.log" 2>&1
echo "model exit=$? $(tail -1 "$OUT/model.log")"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > "$OUT/smoke.log" 2>&1
echo "smoke exit=$? $(tail -1 "$OUT/smoke.log")"
timeout 900 python bench.py --steps 5 --warmup 2 > "$OUT/bench.log" 2>&1
echo "bench exit=$? $(tail -c 600 "$OUT/bench.log")"
