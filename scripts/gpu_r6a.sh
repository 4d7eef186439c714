#!/bin/bash
# Round 6 (a): root-cause experiments for the nondeterministic dswgrad instantiation + the kernels' tests on the padded walk
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/${1:-r6a}
mkdir -p "$OUT"
export SMAAT_REQUIRE_GPU=1
timeout 900 python scripts/probes/r6_dswgrad_rootcause_run.py > "$OUT/dswgrad_rootcause.txt" 2>&1
echo "rootcause exit=$?"; cat "$OUT/dswgrad_rootcause.txt"
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_f16_split.py tests/test_strict_blocks.py tests/test_gpu_bf16.py -q -m gpu --tb=short -p no:cacheprovider -x > "$OUT/pytest_subset.log" 2>&1
echo "pytest exit=$? $(tail -1 "$OUT/pytest_subset.log")"
