#!/bin/bash
# Round 6: full GPU suite + smoke on the tree with the two-term fused forward
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/${1:-r6q}
mkdir -p "$OUT"
export SMAAT_REQUIRE_GPU=1
timeout 2400 python -m pytest tests -x -q -m gpu > "$OUT/pytest_gpu.log" 2>&1
echo "pytest gpu exit=$? $(tail -1 "$OUT/pytest_gpu.log")"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > "$OUT/smoke.log" 2>&1
echo "smoke exit=$? $(tail -1 "$OUT/smoke.log")"
