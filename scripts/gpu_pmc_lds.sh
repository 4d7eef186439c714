#!/bin/bash
# Diagnostic counter passes for the GEMM kernels: where do the cycles of a CU go (LDS active / bank conflicts / waits,
# VALU, VMEM, MFMA)?  usage: gpu_pmc_lds.sh <tag> ["extra bench flags"]
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
REPO=$PWD
export TMPDIR=/tmp
OUT=$REPO/gpurun_out/${1:-pmc_lds}
EXTRA="${2:-}"
mkdir -p "$OUT"
cd /tmp
BENCH1="python $REPO/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-profile --no-alt --no-latency --no-eager-baseline --no-input-pipeline --no-side-configs $EXTRA"
i=0
for pass in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" \
            "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY" \
            "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY" \
            "SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_VMEM"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d "$OUT/pmc_$i" -o bench -- $BENCH1 > "$OUT/pmc_$i.log" 2>&1
  echo "pmc pass $i exit=$?"
done
cd "$REPO"
python - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
disp = collections.defaultdict(int)
for f in glob.glob(out + "/pmc_*/**/*counter_collection.csv", recursive=True):
    seen = set()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        key = (k, r.get("Dispatch_Id"))
        if r["Counter_Name"].startswith(("SQ_LDS_BANK", "SQ_ACTIVE_INST_VALU", "SQ_VALU_MFMA", "SQ_INSTS_LDS")) and key not in seen:
            seen.add(key)
for k in sorted(agg, key=lambda k: -agg[k].get("SQ_BUSY_CU_CYCLES", 0))[:14]:
    c = agg[k]
    print(k[:70])
    print("   " + "  ".join(f"{n.replace('SQ_', '')}={v:.3g}" for n, v in sorted(c.items())))
PY
