#!/bin/bash
# rocprofv3 evidence for one round, all on the SAME bench command:
#   1. --kernel-trace --stats  -> per-kernel time table (copied to profiles/ by scripts/summarize_prof.py)
#   2. --pmc FETCH_SIZE         (own pass)   HBM read bytes per dispatch
#   3. --pmc WRITE_SIZE         (own pass)   HBM write bytes per dispatch
#   4. --pmc SQ_* MFMA counters (own pass)   matrix-pipe busy cycles
# Counter passes never combine with sys/hip/hsa tracing (only --kernel-trace).
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
REPO=$PWD
export TMPDIR=/tmp
OUT=$REPO/gpurun_out/${1:-prof}
EXTRA="${2:-}"   # extra bench.py flags, e.g. "--precision bf16 --batch 64" (BASELINE configs[3])
mkdir -p "$OUT"
cd /tmp
BENCH="python $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-profile --no-alt --no-latency --no-eager-baseline --no-input-pipeline --no-side-configs --no-power $EXTRA"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o bench -- $BENCH > "$OUT/bench_under_rocprof.log" 2>&1
echo "trace exit=$?"
BENCH1="python $REPO/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-profile --no-alt --no-latency --no-eager-baseline --no-input-pipeline --no-side-configs --no-power $EXTRA"
for pass in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE" SQ_INSTS_VALU_MFMA_MOPS_F32; do
  tag=$(echo $pass | cut -d' ' -f1)
  timeout 600 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d "$OUT/pmc_$tag" -o bench -- $BENCH1 > "$OUT/pmc_$tag.log" 2>&1
  echo "pmc $tag exit=$?"
done
rocprofv3 -L > "$OUT/counters.txt" 2>&1
cd "$REPO"
python scripts/summarize_prof.py "$OUT" > "$OUT/summary.txt" 2>&1
echo "summary exit=$?"
tail -40 "$OUT/summary.txt"
# the raw per-dispatch CSVs are large: keep only stats + summaries
find "$OUT" -name "*kernel_trace.csv" -size +8M -delete
find "$OUT" -name "*counter_collection.csv" -size +4M -delete   # (gpurun merges at most 64 MiB back)
find "$OUT" -name "*.db" -delete 2>/dev/null
