#!/bin/bash
# Round 5: BatchNorm backward apply in address order -- kernel tests + interleaved step A/B (SMAAT_BN_LIN=1024 default / 0)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
TAG=${1:-r5n}
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
export SMAAT_REQUIRE_GPU=1
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_bf16.py tests/test_gpu_f16_split.py -q -m gpu -k "bn or amax" --tb=short -p no:cacheprovider > "$OUT/pytest_bn.log" 2>&1
echo "bn tests exit=$? $(tail -1 "$OUT/pytest_bn.log")"
grep -E "^(FAILED|ERROR)|Memory access" "$OUT/pytest_bn.log" | head -30
B="python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-profile --no-alt --no-latency --no-eager-baseline --no-input-pipeline --no-side-configs"
for rep in 1 2 3; do
  for f in 1024 0; do
    SMAAT_BN_LIN=$f timeout 300 $B > "$OUT/bench_bnlin_${f}_$rep.json" 2> "$OUT/bench_bnlin_${f}_$rep.err"
    echo "BN_LIN=$f rep $rep: $(python -c "
import json
try:
    j=json.loads([l for l in open('$OUT/bench_bnlin_${f}_$rep.json') if l.startswith('{')][-1]); print(j['value'], j['ms_per_step'])
except Exception as e: print('parse error', e)
")"
  done
done
