#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
REPO=$PWD
export TMPDIR=/tmp
OUT=$REPO/gpurun_out/${1:-wgbank}
mkdir -p "$OUT"
timeout 300 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "wgrad" --tb=short -p no:cacheprovider > "$OUT/k_wgrad.log" 2>&1
echo "wgrad kernels exit=$? $(tail -1 "$OUT/k_wgrad.log")"
timeout 300 python scripts/layer_bench.py > "$OUT/layer_bench.txt" 2>&1
echo "layer_bench exit=$? $(tail -1 "$OUT/layer_bench.txt")"
cd /tmp
timeout 60 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d "$OUT/sq2" -o wg -- python $REPO/scripts/probes/wgrad_only.py > "$OUT/sq2.log" 2>&1
cd $REPO
python - "$OUT" <<'PY' | tee "$OUT/summary.txt"
import csv, glob, os, sys
from collections import defaultdict
out = sys.argv[1]
for f in sorted(glob.glob(os.path.join(out, "*", "**", "*counter_collection.csv"), recursive=True)):
    acc = defaultdict(lambda: defaultdict(float)); n = defaultdict(set)
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][:48]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k].add(r["Dispatch_Id"])
    for k in acc:
        if "k_wgrad" in k:
            print(f"{k:40s} n={len(n[k]):3d}", "  ".join(f"{c}={v/len(n[k]):.5g}" for c, v in acc[k].items()))
PY
