#!/bin/bash
# round 4: producer micro-optimisations (v_cvt_pk_bf16_f32, zero taps instead of selects) + packed-FMA experiment under
# bf16 storage: kernel parity, per-layer bf16 table with SMAAT_DWG_PK=0/1, bf16 and f32 step
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/${1:-r4i}
mkdir -p "$OUT"
export SMAAT_REQUIRE_GPU=1
for pk in 0 1; do
  SMAAT_DWG_PK=$pk timeout 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu --tb=short -p no:cacheprovider -k "rows or wgrad_split" > "$OUT/pytest_pk$pk.log" 2>&1
  echo "pytest PK=$pk exit=$? $(tail -1 "$OUT/pytest_pk$pk.log")"
  grep -E "^(FAILED|ERROR)|Memory access" "$OUT/pytest_pk$pk.log" | head
  if grep -q "Memory access\|Aborted" "$OUT/pytest_pk$pk.log"; then echo "fault: stop"; exit 1; fi
  for l in inc.1 up4; do SMAAT_DWG_PK=$pk LB_ONLY=$l timeout 600 python scripts/layer_bench_bf16.py >> "$OUT/layer_bench_bf16_pk$pk.txt" 2>&1; done
  echo "PK=$pk"; grep -E "ROWS fwd" "$OUT/layer_bench_bf16_pk$pk.txt" | sed 's/.*| ROWS/ROWS/'
done
for l in inc.1 up4; do LB_ONLY=$l timeout 600 python scripts/layer_bench.py >> "$OUT/layer_bench_f32.txt" 2>&1; done
grep -E "ROWS fwd" "$OUT/layer_bench_f32.txt" | sed 's/.*| ROWS/ROWS/'
for pk in 0 1; do
  SMAAT_DWG_PK=$pk timeout 600 python bench.py --precision bf16 --batch 64 --steps 30 --warmup 5 --no-power > "$OUT/bench_bf16_pk$pk.json" 2> "$OUT/bench_bf16_pk$pk.err"
  echo "bf16 PK=$pk exit=$? $(python -c "
import json
j=json.loads([l for l in open('$OUT/bench_bf16_pk$pk.json') if l.startswith('{')][-1])
k=j['kernels']
print(j['value'], 'f/s', j['ms_per_step'], 'ms', {n:k[n]['ms_per_step'] for n in k if 'wgrad' in n or 'dsconv_fwd' in n or n in ('smaat_dw3x3_fwd_t','smaat_pointwise_fwd_bf16')})
")"
done
