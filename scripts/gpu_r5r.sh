#!/bin/bash
# Round 5, cumulative: the final tree against the same tree with every switch of the round off (round-4 arithmetic and kernels),
# interleaved on one box; then each switch off alone
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
TAG=${1:-r5r}
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
export SMAAT_REQUIRE_GPU=1
B="python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-profile --no-alt --no-latency --no-eager-baseline --no-input-pipeline --no-side-configs"
run() {  # name, env...
  local name=$1; shift
  env "$@" timeout 300 $B > "$OUT/bench_$name.json" 2> "$OUT/bench_$name.err"
  echo "$name: $(python -c "
import json
try:
    j=json.loads([l for l in open('$OUT/bench_$name.json') if l.startswith('{')][-1]); print(j['value'], 'frames/s', j['ms_per_step'], 'ms')
except Exception as e: print('parse error', e)
")"
}
for rep in 1 2; do
  run final_$rep A=1
  run round4_switches_$rep SMAAT_F16_SPLIT=0 SMAAT_CBAM_THREE_PASS=0 SMAAT_BN_LIN=0 SMAAT_DW_LIN=0
done
run no_f16_split SMAAT_F16_SPLIT=0
run no_three_pass SMAAT_CBAM_THREE_PASS=0
run no_bn_lin SMAAT_BN_LIN=0
run no_dw_lin SMAAT_DW_LIN=0
run final_3 A=1
