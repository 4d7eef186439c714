#!/bin/bash
# step time under the two split policies
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/${1:-policy}
mkdir -p "$OUT"
for pol in auto all auto all; do
  SMAAT_SPLIT_POLICY=$pol timeout 400 python bench.py --steps 20 --warmup 4 --no-alt --no-latency --no-cpu-baseline > "$OUT/bench_$pol.json" 2> "$OUT/bench_$pol.err"
  echo "$pol bench: $(python -c "import json,sys; d=json.loads(open('$OUT/bench_$pol.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])")"
done
