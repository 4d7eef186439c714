#!/bin/bash
# Round 6: one-launch Adam -- which contraction variant is torch's, and the time of a step
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/${1:-r6s}
mkdir -p "$OUT"
export SMAAT_REQUIRE_GPU=1
timeout 600 python scripts/probes/adam_variant_probe.py > "$OUT/adam_variant_probe.txt" 2>&1
echo "probe exit=$?"; grep -v amdgpu.ids "$OUT/adam_variant_probe.txt" | tail -14
