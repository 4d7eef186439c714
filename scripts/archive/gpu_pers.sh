#!/bin/bash
# persistent split GEMM (k_pw_split_p): parity + per-layer timing + step time against the one-tile-per-workgroup form
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/${1:-pers}
mkdir -p "$OUT"
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "split" --tb=short -p no:cacheprovider > "$OUT/k_split.log" 2>&1
echo "split kernels exit=$? $(tail -1 "$OUT/k_split.log")"
timeout 900 python -m pytest tests/test_gpu_model.py -q -m gpu -x --tb=short -p no:cacheprovider > "$OUT/model.log" 2>&1
echo "model exit=$? $(tail -1 "$OUT/model.log")"
for cfg in 0 16; do
  SMAAT_PWS_CFG=$cfg timeout 300 python scripts/layer_bench.py > "$OUT/layer_bench_cfg$cfg.txt" 2>&1
  echo "cfg$cfg layer_bench exit=$? $(tail -1 "$OUT/layer_bench_cfg$cfg.txt")"
  SMAAT_PWS_CFG=$cfg timeout 400 python bench.py --steps 10 --warmup 3 --no-alt --no-latency > "$OUT/bench_cfg$cfg.json" 2> "$OUT/bench_cfg$cfg.err"
  echo "cfg$cfg bench: $(python -c "import json,sys; d=json.loads(open('$OUT/bench_cfg$cfg.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])")"
done
