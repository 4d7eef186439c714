#!/bin/bash
# memory-system stall counters of the GEMM-family kernels on ONE layer shape (layer_bench, LB_ONLY)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
REPO=$PWD
export TMPDIR=/tmp
OUT=$REPO/gpurun_out/${1:-pmcm}
mkdir -p "$OUT"
cd /tmp
i=0
for L in ${LAYERS:-up4.0 up2.0}; do
 for pass in "TCC_EA0_WRREQ_STALL TCC_EA0_WRREQ_DRAM_CREDIT_STALL TCC_TOO_MANY_EA_WRREQS_STALL TCC_EA0_WRREQ TCC_EA0_WRREQ_64B TCC_BUSY TCC_CYCLE" \
             "TCC_TAG_STALL TCC_SRC_FIFO_FULL TCC_LATENCY_FIFO_FULL TCC_IB_STALL TCC_REQ TCC_WRITE TCC_READ" \
             "TCP_PENDING_STALL_CYCLES TCP_TCR_TCP_STALL_CYCLES TCP_TCC_WRITE_REQ_LATENCY TCP_TCC_WRITE_REQ TCP_TCC_READ_REQ_LATENCY TCP_TCC_READ_REQ" \
             "SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  LB_ONLY=$L timeout 300 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d "$OUT/${L}_p$i" -o lb -- python $REPO/scripts/layer_bench.py > "$OUT/${L}_p$i.log" 2>&1
  echo "$L pass$i exit=$?"
 done
done
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
        if "pw" in k or "dw3x3" in k:
            print(f"{k:48s} n={len(n[k]):3d}", "  ".join(f"{c}={v/len(n[k]):.4g}" for c, v in acc[k].items()))
PY
