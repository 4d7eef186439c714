#!/bin/bash
# Round 6: which row-walking forward instantiations the step runs with the two-term form on, and what they and the two amax
# producers cost (rocprofv3 kernel statistics over a short bench run), next to the same with it off
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/${1:-r6p}
mkdir -p "$OUT"
export SMAAT_REQUIRE_GPU=1
export TMPDIR=/tmp
for v in 1 0; do
  d=/tmp/prof_h$v
  rm -rf $d
  SMAAT_FWD_ROWS_H=$v rocprofv3 --kernel-trace --stats --output-format csv -d $d -o p -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-alt --no-latency --no-power --no-eager-baseline --no-side-configs --no-input-pipeline --no-profile > "$OUT/bench_prof_h$v.log" 2>&1
  f=$(find $d -name "*kernel_stats.csv" | head -1)
  python - "$f" "$v" > "$OUT/kernels_h$v.txt" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
print("SMAAT_FWD_ROWS_H=%s: kernel time %.3f ms per step (25 steps)"%(sys.argv[2], tot/25e6))
for r in rows:
    n=r["Name"]
    if any(k in n for k in ("k_dsconv_rows_fwd","k_cbam_apply<","k_upsample2x_fwd_rows","k_weight_planes","k_weight_amax","k_pw_split_p<1","k_pwgemm<","k_dw3x3_fwd_rows")):
        print("%6d calls  avg %9.1f us  total/step %7.3f ms  %s"%(int(r["Calls"]), float(r["AverageNs"])/1e3, float(r["TotalDurationNs"])/25e6, n[:110]))
PY
  cat "$OUT/kernels_h$v.txt"
done
