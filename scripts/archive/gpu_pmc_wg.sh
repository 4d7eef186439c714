#!/bin/bash
# SQ/LDS counter passes on the split weight-gradient kernel of one layer (+ split parity tests)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
REPO=$PWD
export TMPDIR=/tmp
OUT=$REPO/gpurun_out/${1:-pmcwg}
mkdir -p "$OUT"
timeout 300 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "split" --tb=short -p no:cacheprovider > "$OUT/k_split.log" 2>&1
echo "split kernels exit=$? $(tail -1 "$OUT/k_split.log")"
timeout 300 python -m pytest tests/test_gpu_model.py -q -m gpu -x --tb=short -p no:cacheprovider > "$OUT/model.log" 2>&1
echo "model exit=$? $(tail -1 "$OUT/model.log")"
cd /tmp
run() {
  tag=$1; shift
  timeout 60 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d "$OUT/$tag" -o wg -- python $REPO/scripts/probes/wgrad_only.py > "$OUT/$tag.log" 2>&1
  echo "$tag exit=$?"
}
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE
run sq2 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD SQ_WAIT_ANY SQ_INSTS_VALU GRBM_GUI_ACTIVE
DG_CIN=128 DG_COUT=64 DG_H=288 run sq1_shallow SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE
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
    print("==", os.path.relpath(f, out))
    for k in acc:
        if "k_wgrad" in k:
            print(f"{k:40s} n={len(n[k]):3d}", "  ".join(f"{c}={v/len(n[k]):.5g}" for c, v in acc[k].items()))
PY
