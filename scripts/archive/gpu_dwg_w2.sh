#!/bin/bash
# round 4: recompute weight gradient under bf16 storage with TWO workgroups per CU (SMAAT_DWG_W2=1) vs one
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/${1:-dwg_w2}
mkdir -p "$OUT"
export SMAAT_REQUIRE_GPU=1
for w2 in 0 1; do
  SMAAT_DWG_W2=$w2 timeout 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu --tb=short -p no:cacheprovider -k "dsconv_wgrad_split" > "$OUT/pytest_w2_$w2.log" 2>&1
  echo "pytest W2=$w2 exit=$? $(tail -1 "$OUT/pytest_w2_$w2.log")"
  for pk in 0 1; do
    echo "== bf16 batch 64 W2=$w2 PK=$pk"
    SMAAT_DWG_W2=$w2 SMAAT_DWG_PK=$pk LB_ONLY=${LB_ONLY:-inc.1,up4.0,up4.1} timeout 300 python scripts/layer_bench_bf16.py 2>&1 | grep -v amdgpu.ids | tee "$OUT/layer_bench_bf16_w2_${w2}_pk$pk.txt" | tail -6
  done
done
echo done
