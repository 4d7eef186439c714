#!/bin/bash
# rocprofv3 kernel-trace stats of the bench + counter list
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/${1:-prof}
mkdir -p "$OUT"
cd /tmp
rocprofv3 --kernel-trace --stats -d "$OUT/trace" -o bench -- python "$GRAFT_REPO_ROOT/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --no-profile > "$OUT/bench_under_rocprof.log" 2>&1
echo "trace exit=$?"
find "$OUT/trace" -name "*stats*" | head
rocprofv3 -L > "$OUT/counters.txt" 2>&1
echo "counters: $(wc -l < "$OUT/counters.txt") lines"
