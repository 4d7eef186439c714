#!/bin/bash
# Counter passes for the recompute weight gradient (k_dsconv_wgrad_split) next to the streamed k_wgrad_split on one layer:
# where do the producer-bound kernel's cycles go (VALU / SALU / LDS conflicts / waits / MFMA)?  usage: gpu_pmc_dswg.sh <tag>
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
REPO=$PWD
export TMPDIR=/tmp
OUT=$REPO/gpurun_out/${1:-pmc_dswg}
mkdir -p "$OUT"
cd /tmp
CMD="python $REPO/scripts/probes/dswgrad_only.py"
i=0
for pass in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" \
            "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY" \
            "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY" \
            "SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU" \
            "SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM GRBM_GUI_ACTIVE SQ_WAVES"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d "$OUT/pmc_$i" -o run -- $CMD > "$OUT/pmc_$i.log" 2>&1
  echo "pmc pass $i exit=$?"
done
cd "$REPO"
python - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob(out + "/pmc_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"]][r["Counter_Name"]] += float(r["Counter_Value"])
with open(out + "/summary.txt", "w") as fh:
    for k in sorted(agg, key=lambda k: -agg[k].get("SQ_BUSY_CU_CYCLES", 0))[:6]:
        c = agg[k]
        line = k[:80] + "\n   " + "  ".join(f"{n.replace('SQ_', '')}={v:.4g}" for n, v in sorted(c.items()))
        print(line)
        fh.write(line + "\n")
PY
