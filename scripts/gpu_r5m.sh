#!/bin/bash
# Round 5: non-temporal hints on the streaming helpers -- interleaved step A/B of the experiment libraries (scripts/build_nt_libs.sh)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
TAG=${1:-r5m}
LIBS=${2:-"base nt1 nt2 nt3"}
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
export SMAAT_REQUIRE_GPU=1
B="python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-profile --no-alt --no-latency --no-eager-baseline --no-input-pipeline --no-side-configs"
for rep in 1 2; do
  for l in $LIBS; do
    if [ "$l" = base ]; then unset SMAAT_LIB; else export SMAAT_LIB=$PWD/smaat_unet_amd/exp/libsmaat_hip_$l.so; fi
    timeout 300 $B > "$OUT/bench_${l}_$rep.json" 2> "$OUT/bench_${l}_$rep.err"
    echo "$l rep $rep: $(python -c "
import json
try:
    j=json.loads([l for l in open('$OUT/bench_${l}_$rep.json') if l.startswith('{')][-1]); print(j['value'], j['ms_per_step'])
except Exception as e: print('parse error', e)
")"
  done
done
unset SMAAT_LIB
