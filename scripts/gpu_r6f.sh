#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/${1:-r6f}
mkdir -p "$OUT"
export SMAAT_REQUIRE_GPU=1
RC_REPS=24 timeout 1200 python scripts/probes/r6_dswgrad_rootcause_run.py > "$OUT/dswgrad_rootcause_positions.txt" 2>&1
echo "rootcause exit=$?"; cat "$OUT/dswgrad_rootcause_positions.txt"
timeout 900 python -m pytest tests/test_gpu_model.py tests/test_eval_and_big.py -q -m gpu --tb=short -p no:cacheprovider -k "64x48" > "$OUT/pytest_tie_masked.log" 2>&1
echo "pytest exit=$? $(tail -3 "$OUT/pytest_tie_masked.log")"
