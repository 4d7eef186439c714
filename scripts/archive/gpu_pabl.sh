#!/bin/bash
# compile-time ablations of k_pw_split_p (smaat_unet_amd/abl/libsmaat_abl<v>.so) on selected layer shapes
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/${1:-pabl}
mkdir -p "$OUT"
for L in ${LAYERS:-up4.0 inc.1 up2.0}; do
  LB_ONLY=$L timeout 200 python scripts/layer_bench.py 2>&1 | grep "^$L" | sed "s/^/base /"
  for v in ${VARIANTS:-1 2 4 3 7}; do
    SMAAT_LIB=$PWD/smaat_unet_amd/abl/libsmaat_abl$v.so LB_ONLY=$L timeout 200 python scripts/layer_bench.py 2>&1 | grep "^$L" | sed "s/^/abl$v /"
  done
done | tee "$OUT/pabl.txt"
