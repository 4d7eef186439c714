#!/bin/bash
# depthwise kernels: correctness subset + per-shape timing, row-streaming vs strip kernels
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/${1:-dw}
mkdir -p "$OUT"
export SMAAT_REQUIRE_GPU=1
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu --tb=short -p no:cacheprovider -k "dw3x3 or dsconv or double_conv or bnred" > "$OUT/pytest.log" 2>&1
echo "pytest exit=$? $(tail -1 "$OUT/pytest.log")"
grep -E "^(FAILED|ERROR)" "$OUT/pytest.log" | head
for lib in ${DW_LIBS:-libsmaat_hip.so}; do
  echo "== $lib"
  SMAAT_LIB=$PWD/smaat_unet_amd/$lib timeout 300 python scripts/dw_bench.py 2>&1 | grep -v amdgpu.ids | tee "$OUT/dw_bench_$lib.txt"
done
echo "== strip kernels"
SMAAT_DW_ROWS=0 timeout 300 python scripts/dw_bench.py 2>&1 | grep -v amdgpu.ids | tee "$OUT/dw_bench_strip.txt" | tail -1
