#!/bin/bash
# Experiment builds of the row-walking kernels with parts of an iteration compiled out (timing only: results are wrong).
#   build_ablation_libs.sh dswgrad "1 2 4 8 16 7 12 15 31"   ->  smaat_unet_amd/exp/libsmaat_hip_dwgdbg<bits>.so  (-DDWG_DBG)
#   build_ablation_libs.sh dsrows  "1 4 8 32 64 96 103 127"  ->  smaat_unet_amd/exp/libsmaat_hip_dsrdbg<bits>.so  (-DDSR_DBG)
# Needs the normal build's objects (make -C smaat_unet_amd/csrc).  Read by scripts/probes/{dswgrad,dsrows}_ablate.py via SMAAT_LIB.
set -eu
cd "$(dirname "$0")/../smaat_unet_amd/csrc"
which=${1:?dswgrad|dsrows}
bits=${2:?list of bit masks}
case $which in dswgrad) mac=DWG_DBG; tag=dwgdbg ;; dsrows) mac=DSR_DBG; tag=dsrdbg ;; *) echo "dswgrad|dsrows"; exit 2 ;; esac
mkdir -p ../exp
others=$(ls *.o | grep -v "^$which.o$" | tr '\n' ' ')
for d in $bits; do
  ( /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -D$mac=$d -c $which.hip -o /tmp/${which}_dbg$d.o 2>/dev/null &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $others /tmp/${which}_dbg$d.o -o ../exp/libsmaat_hip_$tag$d.so ) &
done
wait
ls -la ../exp
