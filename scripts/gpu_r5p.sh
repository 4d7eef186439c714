#!/bin/bash
# Round 5: non-walking depthwise forward -- interleaved step A/B (SMAAT_DW_LIN=2 default policy / 0 walker everywhere)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
TAG=${1:-r5p}
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
export SMAAT_REQUIRE_GPU=1
B="python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-profile --no-alt --no-latency --no-eager-baseline --no-input-pipeline --no-side-configs"
for rep in 1 2 3; do
  for f in 2 0; do
    SMAAT_DW_LIN=$f timeout 300 $B > "$OUT/bench_dwlin_${f}_$rep.json" 2> "$OUT/bench_dwlin_${f}_$rep.err"
    echo "SMAAT_DW_LIN=$f rep $rep: $(python -c "
import json
try:
    j=json.loads([l for l in open('$OUT/bench_dwlin_${f}_$rep.json') if l.startswith('{')][-1]); print(j['value'], 'frames/s', j['ms_per_step'], 'ms')
except Exception as e: print('parse error', e)
")"
  done
done
for p in "--precision bf16 --batch 64"; do
  for f in 2 0; do
    SMAAT_DW_LIN=$f timeout 300 $B $p > "$OUT/bench_bf16_dwlin_${f}.json" 2> "$OUT/bench_bf16_dwlin_${f}.err"
    echo "bf16 b64 SMAAT_DW_LIN=$f: $(python -c "
import json
try:
    j=json.loads([l for l in open('$OUT/bench_bf16_dwlin_${f}.json') if l.startswith('{')][-1]); print(j['value'], 'frames/s', j['ms_per_step'], 'ms')
except Exception as e: print('parse error', e)
")"
  done
done
