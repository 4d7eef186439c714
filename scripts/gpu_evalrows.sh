#!/bin/bash
# round 4: inference half blocks on the row-walking kernel (SMAAT_EVAL_ROWS=1, default) vs the tile kernel (0)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/${1:-evalrows}
mkdir -p "$OUT"
export SMAAT_REQUIRE_GPU=1
timeout 900 python -m pytest tests -q -m gpu --tb=short -p no:cacheprovider -k "eval or folded or graph or latency or infer" > "$OUT/pytest_eval.log" 2>&1
echo "pytest eval exit=$? $(grep -E 'passed|failed' "$OUT/pytest_eval.log" | tail -1)"
grep -E "^(FAILED|ERROR)" "$OUT/pytest_eval.log" | head
for er in 1 0 1 0; do
  echo "== SMAAT_EVAL_ROWS=$er"
  SMAAT_EVAL_ROWS=$er timeout 300 python scripts/eval_latency.py 2>&1 | grep -v amdgpu.ids | tee -a "$OUT/eval_latency_rows$er.txt" | tail -4
done
