#!/bin/bash
# round 4: row-walking kernels after the vmcnt(0) drain fix; SMAAT_DWG_ILV A/B for the recompute weight gradient
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/${1:-rows_fix}
mkdir -p "$OUT"
export SMAAT_REQUIRE_GPU=1
for ilv in 0 1; do
  SMAAT_ROWS_ILV=$ilv timeout 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu --tb=short -p no:cacheprovider -k "dsconv_wgrad_split or dsconv_fwd_rows" > "$OUT/pytest_ilv$ilv.log" 2>&1
  echo "pytest ILV=$ilv exit=$? $(tail -1 "$OUT/pytest_ilv$ilv.log")"
  grep -E "^(FAILED|ERROR)" "$OUT/pytest_ilv$ilv.log" | head -5
  echo "== f32 batch 32 ILV=$ilv"
  SMAAT_ROWS_ILV=$ilv LB_ONLY=inc.1,up4.0,up4.1 timeout 300 python scripts/layer_bench.py 2>&1 | grep -v amdgpu.ids | tee "$OUT/layer_bench_f32_ilv$ilv.txt" | cut -c1-400 | tail -5
  echo "== bf16 batch 64 ILV=$ilv"
  SMAAT_ROWS_ILV=$ilv LB_ONLY=inc.1,up4.0,up4.1 timeout 300 python scripts/layer_bench_bf16.py 2>&1 | grep -v amdgpu.ids | tee "$OUT/layer_bench_bf16_ilv$ilv.txt" | tail -5
done
echo done
