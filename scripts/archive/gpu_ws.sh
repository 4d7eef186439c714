#!/bin/bash
# fused depthwise->pointwise forward (k_pwgemm_ws) after a change: parity + f32-path layer timing + step time
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/${1:-ws}
mkdir -p "$OUT"
timeout 300 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "dsconv_fwd" --tb=short -p no:cacheprovider > "$OUT/k_dsconv_fwd.log" 2>&1
echo "dsconv_fwd exit=$? $(tail -1 "$OUT/k_dsconv_fwd.log")"
timeout 600 python -m pytest tests/test_gpu_model.py -q -m gpu -x --tb=short -p no:cacheprovider > "$OUT/model.log" 2>&1
echo "model exit=$? $(tail -1 "$OUT/model.log")"
SMAAT_SPLIT=0 timeout 300 python scripts/layer_bench.py > "$OUT/layer_bench_f32.txt" 2>&1
echo "layer_bench(f32) exit=$? $(tail -1 "$OUT/layer_bench_f32.txt")"
timeout 400 python bench.py --steps 10 --warmup 3 --no-alt --no-latency --no-cpu-baseline > "$OUT/bench.json" 2> "$OUT/bench.err"
echo "bench: $(python -c "import json,sys; d=json.loads(open('$OUT/bench.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])")"
