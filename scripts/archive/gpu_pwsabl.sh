#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/${1:-pwsabl}
mkdir -p "$OUT"
for abl in 0 1 2 3 5 7; do
  SMAAT_PWS_ABLATE=$abl LB_ONLY="${LB_ONLY:-}" timeout 300 python scripts/layer_bench.py > "$OUT/lb_abl$abl.txt" 2>&1
  echo "abl$abl exit=$? $(tail -1 "$OUT/lb_abl$abl.txt")"
done
