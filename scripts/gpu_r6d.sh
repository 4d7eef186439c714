#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/${1:-r6d}
mkdir -p "$OUT"
export SMAAT_REQUIRE_GPU=1
RC_REPS=32 timeout 900 python scripts/probes/r6_dswgrad_rootcause_run.py > "$OUT/dswgrad_rootcause_regions.txt" 2>&1
echo "rootcause exit=$?"; cat "$OUT/dswgrad_rootcause_regions.txt"
timeout 600 python -m pytest tests/test_gpu_f16_split.py -q -m gpu --tb=short -p no:cacheprovider -k "second_backward or reproducible" > "$OUT/pytest_second_backward.log" 2>&1
echo "pytest exit=$? $(tail -3 "$OUT/pytest_second_backward.log")"
