#!/bin/bash
# weight-gradient split kernel with 8 producer waves (SMAAT_PWS_CFG=8) against the default 4
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/${1:-wg512}
mkdir -p "$OUT"
for cfg in 0 8; do
  SMAAT_PWS_CFG=$cfg timeout 300 python scripts/layer_bench.py > "$OUT/layer_bench_cfg$cfg.txt" 2>&1
  echo "cfg$cfg $(tail -1 "$OUT/layer_bench_cfg$cfg.txt")"
  SMAAT_PWS_CFG=$cfg SMAAT_SPLIT_POLICY=all timeout 400 python bench.py --steps 10 --warmup 3 --no-alt --no-latency --no-cpu-baseline > "$OUT/bench_cfg$cfg.json" 2> "$OUT/bench_cfg$cfg.err"
  echo "cfg$cfg policy=all bench: $(python -c "import json,sys; d=json.loads(open('$OUT/bench_cfg$cfg.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])")"
done
