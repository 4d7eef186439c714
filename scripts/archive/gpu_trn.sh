#!/bin/bash
# transposed-accumulator epilogue of the plain pointwise GEMM: parity + layer timing + step time, on/off
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/${1:-trn}
mkdir -p "$OUT"
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "pointwise" --tb=short -p no:cacheprovider > "$OUT/k_pointwise.log" 2>&1
echo "pointwise exit=$? $(tail -1 "$OUT/k_pointwise.log")"
for abl in 0 32; do
  SMAAT_PW_ABLATE=$abl timeout 300 python scripts/layer_bench.py > "$OUT/layer_bench_abl$abl.txt" 2>&1
  echo "abl$abl layer_bench exit=$?"
  SMAAT_PW_ABLATE=$abl timeout 400 python bench.py --steps 10 --warmup 3 --no-alt --no-latency > "$OUT/bench_abl$abl.json" 2> "$OUT/bench_abl$abl.err"
  echo "abl$abl bench: $(python -c "import json,sys; d=json.loads(open('$OUT/bench_abl$abl.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])")"
done
