#!/bin/bash
# Round 6: the stem's data gradient (64 -> 24 channels at 288^2: the one f32-MFMA GEMM left in the step) on the split GEMM
# (SMAAT_SPLIT_POLICY=all) against the f32 kernel, interleaved
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/${1:-r6y}
mkdir -p "$OUT"
export SMAAT_REQUIRE_GPU=1
for v in auto all auto all; do
  SMAAT_SPLIT_POLICY=$v timeout 600 python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-alt --no-latency --no-power --no-eager-baseline --no-side-configs --no-input-pipeline --no-profile 2>/dev/null | tail -1 > "$OUT/b.json"
  python - "$OUT/b.json" "$v" <<'PY' | tee -a "$OUT/bench_ab_stem_dgrad.txt"
import json,sys
d=json.loads(open(sys.argv[1]).read())
print("SMAAT_SPLIT_POLICY=%-5s %.1f frames/s  %.3f ms/step  final loss %s" % (sys.argv[2], d["value"], d["ms_per_step"], d["config"]["final_loss"]))
PY
done
