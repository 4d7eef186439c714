#!/bin/bash
# round 4, second call: the recompute weight gradient (csrc/dswgrad.hip) -- kernel parity, per-layer table, step A/B --
# plus the two tests that failed in r4a (new reference-derived yardsticks) and the HDF5-fed leg with the native gather
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/${1:-r4b}
mkdir -p "$OUT"
export SMAAT_REQUIRE_GPU=1
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu --tb=short -p no:cacheprovider -k "wgrad" > "$OUT/pytest_wgrad.log" 2>&1
echo "pytest wgrad exit=$? $(tail -1 "$OUT/pytest_wgrad.log")"
grep -E "^(FAILED|ERROR)|rel err" "$OUT/pytest_wgrad.log" | head -20
LB_ONLY=${LB_ONLY:-} timeout 600 python scripts/layer_bench.py > "$OUT/layer_bench.txt" 2>&1
echo "layer bench exit=$?"; grep -E "SPLIT recompute|totals" "$OUT/layer_bench.txt"
timeout 900 python -m pytest tests/test_autocast_yardstick.py tests/test_eval_and_big.py tests/test_gpu_model.py tests/test_strict_blocks.py -q -m gpu --tb=short -p no:cacheprovider > "$OUT/pytest_model.log" 2>&1
echo "pytest model exit=$? $(tail -1 "$OUT/pytest_model.log")"
grep -E "^(FAILED|ERROR)" "$OUT/pytest_model.log" | head -20
for mode in off auto off auto; do
  SMAAT_WGRAD_RECOMPUTE=$mode timeout 600 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-alt --no-latency \
      --no-eager-baseline --no-side-configs --no-input-pipeline --no-power > "$OUT/bench_$mode.json" 2> "$OUT/bench_$mode.err"
  echo "bench recompute=$mode exit=$? $(python -c "
import json
j=json.loads([l for l in open('$OUT/bench_$mode.json') if l.startswith('{')][-1])
k=j['kernels']
print(j['value'], 'f/s', j['ms_per_step'], 'ms', {n:k[n]['ms_per_step'] for n in k if 'wgrad' in n or 'dsconv_fwd' in n or n=='smaat_dw3x3_fwd'})
")"
done
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-alt --no-latency --no-eager-baseline --no-side-configs \
    --no-profile --no-power > "$OUT/bench_fed.json" 2> "$OUT/bench_fed.err"
echo "bench fed exit=$? $(python -c "
import json
j=json.loads([l for l in open('$OUT/bench_fed.json') if l.startswith('{')][-1])
print(j['value'], 'fed', j['input_pipeline_fed'].get('value'), 'fed_h5', {k:v for k,v in j['input_pipeline_fed_hdf5'].items() if k!='what'})
")"
