#!/bin/bash
# A/B of the fused forward producer variants (libraries under smaat_unet_amd/abl) on the plane-dominated layers
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/${1:-wsab}
mkdir -p "$OUT"
for v in old d1 d2 d3; do
  lib=$PWD/smaat_unet_amd/abl/libsmaat_ws_$v.so
  [ $v = d3 ] && lib=$PWD/smaat_unet_amd/libsmaat_hip.so
  for L in inc.1 up4.0 up3.1; do
    SMAAT_LIB=$lib SMAAT_SPLIT=0 LB_ONLY=$L timeout 100 python scripts/layer_bench.py 2>&1 | grep "^$L" | cut -c1-80 | sed "s/^/$v /"
  done
  SMAAT_LIB=$lib timeout 300 python bench.py --steps 8 --warmup 3 --no-alt --no-latency --no-cpu-baseline > "$OUT/bench_$v.json" 2> "$OUT/bench_$v.err"
  echo "$v bench: $(python -c "import json,sys; d=json.loads(open('$OUT/bench_$v.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])")"
done | tee "$OUT/ab.txt"
