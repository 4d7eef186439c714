#!/bin/bash
# whole GPU suite in one process (what the driver runs at round end) + smoke; logs under gpurun_out/<tag>/
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/${1:-tests}
mkdir -p "$OUT"
export SMAAT_REQUIRE_GPU=1
timeout ${2:-900} python -m pytest tests -q -m gpu --tb=short -p no:cacheprovider ${3:-} > "$OUT/pytest_gpu.log" 2>&1
echo "pytest exit=$? $(tail -1 "$OUT/pytest_gpu.log")"
grep -E "^(FAILED|ERROR)" "$OUT/pytest_gpu.log" | head -40
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > "$OUT/smoke.log" 2>&1
echo "smoke exit=$? $(tail -1 "$OUT/smoke.log")"
