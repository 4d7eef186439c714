#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/${1:-r6v}
mkdir -p "$OUT"
export SMAAT_REQUIRE_GPU=1
timeout 600 python scripts/probes/host_cost_probe.py > "$OUT/host_cost_probe.txt" 2>&1
grep -v amdgpu.ids "$OUT/host_cost_probe.txt" | tail -6
