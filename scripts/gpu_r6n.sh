#!/bin/bash
# Round 6: fused row-walking forward on the two-term fp16 split with an a-priori |y| bound -- error and time per layer
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/${1:-r6n}
mkdir -p "$OUT"
export SMAAT_REQUIRE_GPU=1
timeout 600 python scripts/probes/rows_fwd_h_probe.py > "$OUT/rows_fwd_h_probe.txt" 2>&1
echo "probe exit=$?"; cat "$OUT/rows_fwd_h_probe.txt" | tail -12
