#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/${1:-r6k}
mkdir -p "$OUT"
export SMAAT_REQUIRE_GPU=1
RC_REPS=24 timeout 1500 python scripts/probes/r6_dswgrad_rootcause_run.py > "$OUT/dswgrad_rootcause_positions_valu.txt" 2>&1
echo "rootcause exit=$?"; cat "$OUT/dswgrad_rootcause_positions_valu.txt"
