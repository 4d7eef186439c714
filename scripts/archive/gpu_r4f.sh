#!/bin/bash
# round 4: row-walking fused forward (csrc/dsrows.hip) -- kernel parity (f32 + bf16 storage), per-layer table, step A/B
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/${1:-r4f}
mkdir -p "$OUT"
export SMAAT_REQUIRE_GPU=1
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu --tb=short -p no:cacheprovider -k "rows or wgrad_split" > "$OUT/pytest_rows.log" 2>&1
echo "pytest rows exit=$? $(tail -1 "$OUT/pytest_rows.log")"
grep -E "^(FAILED|ERROR)|rel err|Memory access" "$OUT/pytest_rows.log" | head -20
if grep -q "Memory access\|Aborted" "$OUT/pytest_rows.log"; then echo "fault: stop"; exit 1; fi
for l in inc.1 up4; do LB_ONLY=$l timeout 600 python scripts/layer_bench.py >> "$OUT/layer_bench.txt" 2>&1; done
grep -E "ROWS fwd" "$OUT/layer_bench.txt" | sed 's/.*FUSED/FUSED/'
if ! grep -q " passed" "$OUT/pytest_rows.log" || grep -q "failed" "$OUT/pytest_rows.log"; then echo "parity not green: no step A/B"; exit 1; fi
timeout 900 python -m pytest tests/test_gpu_model.py tests/test_eval_and_big.py tests/test_strict_blocks.py -q -m gpu --tb=short -p no:cacheprovider -x > "$OUT/pytest_model.log" 2>&1
echo "pytest model exit=$? $(tail -1 "$OUT/pytest_model.log")"
grep -E "^(FAILED|ERROR)" "$OUT/pytest_model.log" | head
for mode in off auto off auto; do
  SMAAT_FWD_ROWS=$mode timeout 600 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-alt --no-latency \
      --no-eager-baseline --no-side-configs --no-input-pipeline --no-power > "$OUT/bench_rows_$mode.json" 2> "$OUT/bench_rows_$mode.err"
  echo "bench fwd_rows=$mode exit=$? $(python -c "
import json
j=json.loads([l for l in open('$OUT/bench_rows_$mode.json') if l.startswith('{')][-1])
k=j['kernels']
print(j['value'], 'f/s', j['ms_per_step'], 'ms', {n:k[n]['ms_per_step'] for n in k if 'wgrad' in n or 'dsconv_fwd' in n or n=='smaat_dw3x3_fwd'})
")"
done
