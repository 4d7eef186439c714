#!/bin/bash
# Round 5: three-pass attention backward -- kernel tests, model-level tests, interleaved step A/B, kernel statistics
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
TAG=${1:-r5h}
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
export SMAAT_REQUIRE_GPU=1
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_bf16.py -q -m gpu -k "cbam" --tb=short -p no:cacheprovider > "$OUT/pytest_cbam.log" 2>&1
echo "cbam tests exit=$? $(tail -1 "$OUT/pytest_cbam.log")"
grep -E "^(FAILED|ERROR)|Memory access" "$OUT/pytest_cbam.log" | head -30
SMAAT_CBAM_CS=1 timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_bf16.py -q -m gpu -k "three_pass" --tb=short -p no:cacheprovider > "$OUT/pytest_cbam_cs1.log" 2>&1
echo "three-pass tests with one wave per block (bit-exact dbn) exit=$? $(tail -1 "$OUT/pytest_cbam_cs1.log")"
SMAAT_CBAM_CS=8 timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_bf16.py -q -m gpu -k "three_pass" --tb=short -p no:cacheprovider > "$OUT/pytest_cbam_cs8.log" 2>&1
echo "three-pass tests with eight waves per block exit=$? $(tail -1 "$OUT/pytest_cbam_cs8.log")"
timeout 1200 python -m pytest tests/test_gpu_model.py tests/test_strict_blocks.py -q -m gpu --tb=short -p no:cacheprovider > "$OUT/pytest_model.log" 2>&1
echo "model tests exit=$? $(tail -1 "$OUT/pytest_model.log")"
grep -E "^(FAILED|ERROR)|Memory access" "$OUT/pytest_model.log" | head -30
B="python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-profile --no-alt --no-latency --no-eager-baseline --no-input-pipeline --no-side-configs"
for rep in 1 2; do
  for f in 1 0; do
    SMAAT_CBAM_THREE_PASS=$f timeout 300 $B > "$OUT/bench_three_${f}_$rep.json" 2> "$OUT/bench_three_${f}_$rep.err"
    echo "THREE_PASS=$f rep $rep: $(python -c "
import json
try:
    j=json.loads([l for l in open('$OUT/bench_three_${f}_$rep.json') if l.startswith('{')][-1]); print(j['value'], j['ms_per_step'])
except Exception as e: print('parse error', e)
")"
  done
done
cd /tmp && export TMPDIR=/tmp
R="${GRAFT_REPO_ROOT:-/root/repo}"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/$OUT/trace" -o bench -- python "$R/bench.py" --steps 4 --warmup 2 --no-cpu-baseline --no-profile --no-alt --no-latency --no-eager-baseline --no-input-pipeline --no-side-configs > "$R/$OUT/trace.log" 2>&1
echo "rocprof exit=$?"
cd "$R"
F=$(find "$OUT/trace" -name "*kernel_stats.csv" | head -1)
[ -n "$F" ] && { cp "$F" "$OUT/kernel_stats.csv"; grep -i "cbam" "$F" | awk -F, '{printf "%-70s calls %s total_us %.1f avg_us %.1f\n", substr($1,1,70), $2, $3/1000, $4/1000}'; }
find "$OUT/trace" -name "*.csv" ! -name "*kernel_stats.csv" -delete 2>/dev/null
