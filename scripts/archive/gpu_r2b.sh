#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/${1:-r2b}
mkdir -p "$OUT"
bash scripts/gpu_tests.sh "$1" 800
timeout 900 python bench.py > "$OUT/bench.log" 2> "$OUT/bench.err"
echo "bench exit=$? $(head -c 1500 "$OUT/bench.log")"
