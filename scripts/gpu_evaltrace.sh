#!/bin/bash
# kernel list of the batch-1 eval forward (rocprofv3 --kernel-trace --stats over scripts/eval_latency.py): per-kernel
# calls / average duration, to see which launches make up the 0.67 ms.  usage: gpu_evaltrace.sh <tag>
cd "${GRAFT_REPO_ROOT:-/root/repo}"
REPO=$PWD
export TMPDIR=/tmp
OUT=$REPO/gpurun_out/${1:-evaltrace}
mkdir -p "$OUT"
cd /tmp
ITERS=40 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o ev -- python $REPO/scripts/eval_latency.py > "$OUT/eval_under_rocprof.log" 2>&1
echo "trace exit=$?"
cd $REPO
tail -1 "$OUT/eval_under_rocprof.log" | cut -c1-300
python - "$OUT" <<'PY'
import csv, sys, glob
f = glob.glob(sys.argv[1] + "/trace/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
out = []
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:32]:
    out.append(f'{100 * float(r["TotalDurationNs"]) / tot:6.2f}%  calls {int(r["Calls"]):5d}  avg {float(r["AverageNs"]) / 1e3:8.2f} us  {r["Name"][:110]}')
open(sys.argv[1] + "/eval_kernel_table.txt", "w").write("\n".join(out) + "\n")
print("\n".join(out))
PY
