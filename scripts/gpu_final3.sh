#!/bin/bash
# whole GPU suite + smoke, then gpu_final2.sh (default bench line, rocprofv3 statistics + counter passes, traffic record, 2-rank pre-flight)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
TAG=${1:-final3}
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
export SMAAT_REQUIRE_GPU=1
timeout 2400 python -m pytest tests -x -q -m gpu > "$OUT/pytest_gpu.log" 2>&1
echo "pytest gpu exit=$? $(grep -E 'passed|failed' "$OUT/pytest_gpu.log" | tail -1)"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > "$OUT/smoke.log" 2>&1
echo "smoke exit=$? $(tail -1 "$OUT/smoke.log")"
bash scripts/gpu_final2.sh "$TAG"
