#!/bin/bash
# split GEMM after a change: parity + per-layer timing + step time (+ optional ablation library on selected layers)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/${1:-pers2}
mkdir -p "$OUT"
timeout 300 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "split or wgrad" --tb=short -p no:cacheprovider > "$OUT/k_split.log" 2>&1
echo "split kernels exit=$? $(tail -1 "$OUT/k_split.log")"
timeout 600 python -m pytest tests/test_gpu_model.py -q -m gpu -x --tb=short -p no:cacheprovider > "$OUT/model.log" 2>&1
echo "model exit=$? $(tail -1 "$OUT/model.log")"
timeout 300 python scripts/layer_bench.py > "$OUT/layer_bench.txt" 2>&1
echo "layer_bench exit=$? $(tail -1 "$OUT/layer_bench.txt")"
timeout 400 python bench.py --steps 10 --warmup 3 --no-alt --no-latency > "$OUT/bench.json" 2> "$OUT/bench.err"
echo "bench: $(python -c "import json,sys; d=json.loads(open('$OUT/bench.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])")"
