#!/bin/bash
# round 4: recompute weight gradient with the items of an XCD dealt round-robin to its workgroups (SMAAT_DWG_ILV=1)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/${1:-dwg_ilv}
mkdir -p "$OUT"
export SMAAT_REQUIRE_GPU=1
for ilv in 0 1; do
  SMAAT_DWG_ILV=$ilv timeout 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu --tb=short -p no:cacheprovider -k "dsconv_wgrad_split" > "$OUT/pytest_ilv$ilv.log" 2>&1
  echo "pytest ILV=$ilv exit=$? $(tail -1 "$OUT/pytest_ilv$ilv.log")"
  echo "== ILV=$ilv Cin=64"
  SMAAT_DWG_ILV=$ilv DWG_DBGS=0,7,8,0 timeout 300 python scripts/probes/dswgrad_ablate.py 2>&1 | grep -v amdgpu.ids | tee "$OUT/ablate_ilv$ilv.txt"
  echo "== ILV=$ilv Cin=128"
  SMAAT_DWG_ILV=$ilv DG_CIN=128 DWG_DBGS=0,7 timeout 300 python scripts/probes/dswgrad_ablate.py 2>&1 | grep -v amdgpu.ids | tee "$OUT/ablate_cin128_ilv$ilv.txt"
done
echo done
