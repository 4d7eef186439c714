#!/bin/bash
# small counter passes on the split data-gradient GEMM of one layer (base library and an ablation library)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
REPO=$PWD
export TMPDIR=/tmp
OUT=$REPO/gpurun_out/${1:-pmcdg}
mkdir -p "$OUT"
cd /tmp
run() {  # tag lib counters...
  tag=$1; lib=$2; shift 2
  SMAAT_LIB=$lib timeout 60 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d "$OUT/$tag" -o dg -- python $REPO/scripts/probes/dgrad_only.py > "$OUT/$tag.log" 2>&1
  echo "$tag exit=$?"
}
BASE=$REPO/smaat_unet_amd/libsmaat_hip.so
ABL1=$REPO/smaat_unet_amd/abl/libsmaat_abl1.so
run lat_base $BASE TCP_TCC_READ_REQ_LATENCY TCP_TCC_READ_REQ TCP_TCC_WRITE_REQ_LATENCY TCP_TCC_WRITE_REQ
run lat_abl1 $ABL1 TCP_TCC_READ_REQ_LATENCY TCP_TCC_READ_REQ TCP_TCC_WRITE_REQ_LATENCY TCP_TCC_WRITE_REQ
run sq_base $BASE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD GRBM_GUI_ACTIVE
run fifo_base $BASE SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL GRBM_GUI_ACTIVE
run tcp_base $BASE TCP_PENDING_STALL_CYCLES TCP_TCR_TCP_STALL_CYCLES TCP_TCP_TA_DATA_STALL_CYCLES GRBM_GUI_ACTIVE
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
        if "k_pw_split" in k:
            print(f"{k:40s} n={len(n[k]):3d}", "  ".join(f"{c}={v/len(n[k]):.5g}" for c, v in acc[k].items()))
PY
