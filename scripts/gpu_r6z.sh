#!/bin/bash
# Round 6: prototype GEMM on pre-split fp16 planes against the shipped two-term GEMM, per layer shape
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/${1:-r6z}
mkdir -p "$OUT"
export SMAAT_REQUIRE_GPU=1
timeout 300 python scripts/probes/h2_gemm_probe.py > "$OUT/h2_gemm_probe.txt" 2>&1
echo "probe exit=$?"; grep -v amdgpu.ids "$OUT/h2_gemm_probe.txt" | tail -20 | cut -c1-330
