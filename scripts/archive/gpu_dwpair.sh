#!/bin/bash
# experiment: depthwise forward with rows fetched in pairs (-DDWR_FWD_PAIR=1, built as libsmaat_hip_pair.so) vs the default
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/${1:-dwpair}
mkdir -p "$OUT"
export SMAAT_REQUIRE_GPU=1
SMAAT_LIB=$PWD/smaat_unet_amd/libsmaat_hip_pair.so timeout 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu --tb=short -p no:cacheprovider -k "dw3x3_fwd" > "$OUT/pytest_pair.log" 2>&1
echo "pytest (pair lib) exit=$? $(tail -1 "$OUT/pytest_pair.log")"
for rep in 1 2; do
for lib in libsmaat_hip.so libsmaat_hip_pair.so; do
  echo "== $lib (rep $rep)"
  SMAAT_LIB=$PWD/smaat_unet_amd/$lib timeout 300 python scripts/dw_bench.py 2>&1 | grep -v amdgpu.ids | tee -a "$OUT/dw_bench_$lib.txt" | awk '{print $1, $2, $3, $4, $5, $6, $7}'
done
done
