#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/${1:-r6i}
mkdir -p "$OUT"
export SMAAT_REQUIRE_GPU=1
timeout 1200 python scripts/probes/dsbwd_ablate.py > "$OUT/dsbwd_ablate.txt" 2>&1
cat "$OUT/dsbwd_ablate.txt"
