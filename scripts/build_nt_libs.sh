#!/bin/bash
# Experiment builds of the whole library with the non-temporal hint on the streaming helpers of csrc/common.h:
#   build_nt_libs.sh "1 2 3"   ->  smaat_unet_amd/exp/libsmaat_hip_nt<bits>.so   (-DSMAAT_NT=<bits>; read through SMAAT_LIB)
set -eu
cd "$(dirname "$0")/../smaat_unet_amd/csrc"
bits=${1:?list of bit masks}
mkdir -p ../exp
srcs=$(sed -n 's/^SRCS = //p' Makefile)
for d in $bits; do
  (
    tmp=/tmp/smaat_nt$d; mkdir -p $tmp
    for s in $srcs; do
      /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -DSMAAT_NT=$d -c $s -o $tmp/${s%.hip}.o 2>/dev/null &
    done
    wait
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $tmp/*.o -o ../exp/libsmaat_hip_nt$d.so
  ) &
done
wait
ls -la ../exp | grep nt
