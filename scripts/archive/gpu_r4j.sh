#!/bin/bash
# round 4: bf16-storage variants with eight rows of loads in flight -- kernel parity, per-layer bf16 table, bf16 step A/B
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/${1:-r4j}
mkdir -p "$OUT"
export SMAAT_REQUIRE_GPU=1
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu --tb=short -p no:cacheprovider -k "rows or wgrad_split" > "$OUT/pytest.log" 2>&1
echo "pytest exit=$? $(tail -1 "$OUT/pytest.log")"
grep -E "^(FAILED|ERROR)|Memory access" "$OUT/pytest.log" | head
if grep -q "Memory access\|Aborted\|failed" "$OUT/pytest.log"; then echo "not green: stop"; exit 1; fi
for l in inc.1 up4; do LB_ONLY=$l timeout 600 python scripts/layer_bench_bf16.py >> "$OUT/layer_bench_bf16.txt" 2>&1; done
grep -E "ROWS fwd" "$OUT/layer_bench_bf16.txt" | sed 's/.*| ROWS/ROWS/'
for mode in 0 1 0 1; do
  SMAAT_BF16_RECOMPUTE=$mode timeout 600 python bench.py --precision bf16 --batch 64 --steps 30 --warmup 5 --no-power > "$OUT/bench_bf16_rc$mode.json" 2> "$OUT/bench_bf16_rc$mode.err"
  echo "bf16 recompute=$mode exit=$? $(python -c "
import json
j=json.loads([l for l in open('$OUT/bench_bf16_rc$mode.json') if l.startswith('{')][-1])
k=j['kernels']
print(j['value'], 'f/s', j['ms_per_step'], 'ms', {n:k[n]['ms_per_step'] for n in k if 'wgrad' in n or 'dsconv_fwd' in n or n in ('smaat_dw3x3_fwd_t','smaat_pointwise_fwd_bf16')})
")"
done
