#!/bin/bash
# Round 6: 150-step training soak with the one-launch Adam (two runs bit-identical), and with torch's foreach Adam beside it
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/${1:-r6w}
mkdir -p "$OUT"
export SMAAT_REQUIRE_GPU=1
SMAAT_ADAM=one timeout 900 python scripts/probes/soak_determinism.py > "$OUT/soak_determinism_adam_one.txt" 2>&1
echo "one exit=$? $(tail -1 "$OUT/soak_determinism_adam_one.txt")"
SMAAT_ADAM=foreach timeout 900 python scripts/probes/soak_determinism.py > "$OUT/soak_determinism_adam_foreach.txt" 2>&1
echo "foreach exit=$? $(tail -1 "$OUT/soak_determinism_adam_foreach.txt")"
