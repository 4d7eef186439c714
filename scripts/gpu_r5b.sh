#!/bin/bash
# Round 5, second call: whole GPU suite on the policy build (fp16 split where N*H*W >= 4096; amax without a tail round trip),
# interleaved step A/B of the split and of torch's fused Adam, the op-level profile of a step, a profiled default-size bench.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
TAG=${1:-r5b}
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
export SMAAT_REQUIRE_GPU=1
timeout 1500 python -m pytest tests -q -m gpu --tb=short -p no:cacheprovider --durations=8 > "$OUT/pytest_gpu.log" 2>&1
echo "suite exit=$? $(tail -1 "$OUT/pytest_gpu.log")"
grep -E "^(FAILED|ERROR)|Memory access" "$OUT/pytest_gpu.log" | head -30
B="python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-profile --no-alt --no-latency --no-eager-baseline --no-input-pipeline --no-side-configs"
run() { # name, env...
  local nm=$1; shift
  env "$@" timeout 300 $B > "$OUT/bench_$nm.json" 2> "$OUT/bench_$nm.err"
  echo "$nm: $(python -c "
import json
try:
    j=json.loads([l for l in open('$OUT/bench_$nm.json') if l.startswith('{')][-1]); print(j['value'], j['ms_per_step'])
except Exception as e: print('parse error', e)
")"
}
for rep in 1 2; do
  run f16on_$rep SMAAT_F16_SPLIT=1
  run f16off_$rep SMAAT_F16_SPLIT=0
  run adamfused_$rep SMAAT_ADAM=fused
done
timeout 300 python scripts/probes/step_ops_profile.py > "$OUT/step_ops.txt" 2>&1
echo "step ops exit=$?"
timeout 600 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-alt --no-latency --no-eager-baseline --no-input-pipeline --no-side-configs > "$OUT/bench_profiled.json" 2> "$OUT/bench_profiled.err"
echo "profiled bench exit=$?"
python - <<PY
import json
j=json.loads([l for l in open("$OUT/bench_profiled.json") if l.startswith("{")][-1])
print(j["value"], j["ms_per_step"])
for n,d in list(j["kernels"].items())[:14]:
    print(f"{n:36s} {d['calls']:4d} {d['ms_per_step']:7.3f} {d.get('tflops','')} {d.get('alg_gbs','')}")
PY
