#!/bin/bash
# round 4: mixed precision (bf16 storage) on the row-walking forward + typed recompute weight gradient -- model parity,
# bf16 step A/B (SMAAT_BF16_RECOMPUTE), f32 step with the final policy, bf16 per-layer numbers
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/${1:-r4h}
mkdir -p "$OUT"
export SMAAT_REQUIRE_GPU=1
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_bf16.py tests/test_gpu_model.py tests/test_autocast_yardstick.py tests/test_eval_and_big.py \
    -q -m gpu --tb=short -p no:cacheprovider > "$OUT/pytest.log" 2>&1
echo "pytest exit=$? $(tail -1 "$OUT/pytest.log")"
grep -E "^(FAILED|ERROR)|Memory access" "$OUT/pytest.log" | head -20
if grep -q "Memory access\|Aborted" "$OUT/pytest.log"; then echo "fault: stop"; exit 1; fi
for mode in 0 1 0 1; do
  SMAAT_BF16_RECOMPUTE=$mode timeout 600 python bench.py --precision bf16 --batch 64 --steps 30 --warmup 5 --no-power > "$OUT/bench_bf16_rc$mode.json" 2> "$OUT/bench_bf16_rc$mode.err"
  echo "bf16 recompute=$mode exit=$? $(python -c "
import json
j=json.loads([l for l in open('$OUT/bench_bf16_rc$mode.json') if l.startswith('{')][-1])
k=j['kernels']
print(j['value'], 'f/s', j['ms_per_step'], 'ms', {n:k[n]['ms_per_step'] for n in k if 'wgrad' in n or 'dsconv_fwd' in n or n in ('smaat_dw3x3_fwd_t','smaat_pointwise_fwd_bf16')})
")"
done
timeout 600 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-alt --no-latency --no-eager-baseline --no-side-configs \
    --no-input-pipeline --no-power > "$OUT/bench_f32.json" 2> "$OUT/bench_f32.err"
echo "f32 exit=$? $(python -c "
import json
j=json.loads([l for l in open('$OUT/bench_f32.json') if l.startswith('{')][-1])
k=j['kernels']
print(j['value'], 'f/s', j['ms_per_step'], 'ms', {n:k[n]['ms_per_step'] for n in k if 'wgrad' in n or 'dsconv_fwd' in n or n=='smaat_dw3x3_fwd'})
")"
