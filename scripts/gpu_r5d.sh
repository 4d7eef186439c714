#!/bin/bash
# Round 5: fp16 recompute weight gradient -- kernel tests, per-layer table, whole suite, interleaved step A/B
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
TAG=${1:-r5d}
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
export SMAAT_REQUIRE_GPU=1
timeout 900 python -m pytest tests/test_gpu_f16_split.py tests/test_strict_blocks.py -q -m gpu --tb=short -p no:cacheprovider > "$OUT/pytest_f16.log" 2>&1
echo "f16 tests exit=$? $(tail -1 "$OUT/pytest_f16.log")"
grep -E "^(FAILED|ERROR)|Memory access" "$OUT/pytest_f16.log" | head -30
LB_ONLY=inc.1,up4 timeout 600 python scripts/layer_bench_f16.py > "$OUT/layer_bench_f16.txt" 2>&1
echo "layer bench exit=$?"; grep -v amdgpu.ids "$OUT/layer_bench_f16.txt" | tail -12
timeout 1500 python -m pytest tests -q -m gpu --tb=short -p no:cacheprovider --durations=8 > "$OUT/pytest_gpu.log" 2>&1
echo "suite exit=$? $(tail -1 "$OUT/pytest_gpu.log")"
grep -E "^(FAILED|ERROR)|Memory access" "$OUT/pytest_gpu.log" | head -30
grep -E "passed|failed" "$OUT/pytest_gpu.log" | tail -2
B="python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-profile --no-alt --no-latency --no-eager-baseline --no-input-pipeline --no-side-configs"
for rep in 1 2; do
  for f in 1 0; do
    SMAAT_F16_SPLIT=$f timeout 300 $B > "$OUT/bench_f16_${f}_$rep.json" 2> "$OUT/bench_f16_${f}_$rep.err"
    echo "F16_SPLIT=$f rep $rep: $(python -c "
import json
try:
    j=json.loads([l for l in open('$OUT/bench_f16_${f}_$rep.json') if l.startswith('{')][-1]); print(j['value'], j['ms_per_step'])
except Exception as e: print('parse error', e)
")"
  done
done
