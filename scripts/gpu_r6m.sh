#!/bin/bash
# Round 6: end-to-end determinism soak of the final tree (two identical 150-step training runs must agree bit for bit), default
# policy and with the fused backward selected
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/${1:-r6m}
mkdir -p "$OUT"
export SMAAT_REQUIRE_GPU=1
timeout 900 python scripts/probes/soak_determinism.py > "$OUT/soak_determinism.txt" 2>&1
echo "soak default exit=$? $(tail -1 "$OUT/soak_determinism.txt")"
SMAAT_FUSED_BWD=1 SOAK_STEPS=60 timeout 900 python scripts/probes/soak_determinism.py > "$OUT/soak_determinism_fused_bwd.txt" 2>&1
echo "soak fused-backward exit=$? $(tail -1 "$OUT/soak_determinism_fused_bwd.txt")"
