#!/bin/bash
# PMC counters of the GEMM-family kernels on ONE layer shape (layer_bench, LB_ONLY)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
REPO=$PWD
export TMPDIR=/tmp
OUT=$REPO/gpurun_out/${1:-pmcl}
mkdir -p "$OUT"
cd /tmp
for L in ${LAYERS:-up4.0 up2.0}; do
 for pass in "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VMEM GRBM_GUI_ACTIVE"; do
  tag=$(echo $pass | cut -d' ' -f1)
  LB_ONLY=$L LB_ITERS=2 timeout 300 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d "$OUT/${L}_$tag" -o lb -- python $REPO/scripts/layer_bench.py > "$OUT/${L}_$tag.log" 2>&1
  echo "$L $tag exit=$?"
 done
done
cd $REPO
python - "$OUT" <<'PY'
import csv, glob, os, sys
from collections import defaultdict
out = sys.argv[1]
for f in sorted(glob.glob(os.path.join(out, "*", "**", "*counter_collection.csv"), recursive=True)):
    acc = defaultdict(lambda: defaultdict(float)); n = defaultdict(int)
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][:60]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    print("==", os.path.relpath(f, out))
    for k in acc:
        if "pwgemm" in k or "wgrad" in k or "dw3x3" in k:
            print(f"{k:60s}", "  ".join(f"{c}={v:.4g}" for c, v in acc[k].items()))
PY
find "$OUT" -name "*kernel_trace.csv" -delete
