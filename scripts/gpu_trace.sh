#!/bin/bash
# kernel-trace statistics of the bench step only (no counter passes): where the step time goes
cd "${GRAFT_REPO_ROOT:-/root/repo}"
REPO=$PWD
export TMPDIR=/tmp
OUT=$REPO/gpurun_out/${1:-trace}
mkdir -p "$OUT"
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o bench -- python $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-profile --no-alt --no-latency > "$OUT/bench_under_rocprof.log" 2>&1
echo "trace exit=$?"
cd $REPO
python scripts/summarize_prof.py "$OUT" > "$OUT/summary.txt" 2>&1
head -45 "$OUT/summary.txt"
find "$OUT" -name "*kernel_trace.csv" -size +8M -delete
