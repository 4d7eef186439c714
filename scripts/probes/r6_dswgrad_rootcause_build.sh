#!/bin/bash
# Round 6: experiment builds that discriminate between the candidate causes of the call-to-call nondeterminism of
# k_dsconv_wgrad_split<NT=2, AFF, scalar math>.  Every build forces that instantiation into the launcher.
#   old_*  = the round-5 source (git 8d59a1d) with one change each;  new_* = the round-6 source (unconditional padded walk)
# Libraries land in smaat_unet_amd/exp/ (git-ignored, travel with gpurun); scripts/probes/r6_dswgrad_rootcause_run.py runs them.
set -eu
cd "$(dirname "$0")/../../smaat_unet_amd/csrc"
mkdir -p ../exp /tmp/r6rc
OLD=/tmp/r6rc/old.hip
git show 8d59a1d:smaat_unet_amd/csrc/dswgrad.hip > $OLD
others=$(ls *.o | grep -v "^dswgrad.o$" | tr '\n' ' ')
force='s/if (aff) return launch_dswg_cfg<2, true, true, float, float>(a, st);/if (aff) return launch_dswg_cfg<2, true, false, float, float>(a, st);/'
mk() {  # tag, source, extra flags...
  local tag=$1 src=$2; shift 2
  cp $src /tmp/r6rc/dswgrad_$tag.hip
  ( cd /tmp/r6rc && /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -I"$OLDPWD" -I"$OLDPWD/../../include" "$@" -c dswgrad_$tag.hip -o dswgrad_$tag.o 2>/dev/null ) &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $others /tmp/r6rc/dswgrad_$tag.o -o ../exp/libsmaat_hip_rc_$tag.so && echo "built $tag"
}
sed "$force" $OLD > /tmp/r6rc/s_old_scalar.hip
mk old_scalar /tmp/r6rc/s_old_scalar.hip &
# compiler-inserted waits all forced to zero (the asm waits stay counted): cures it <=> the compiler's own wait accounting
mk old_scalar_forcezero /tmp/r6rc/s_old_scalar.hip -mllvm -amdgpu-waitcnt-forcezero=1 &
# the hand-counted wait replaced by a full drain: cures it <=> a prefetched register is consumed before its load landed
sed "$force; s/\"n\"((PD - 1) \* LPG)/\"n\"(0)/" $OLD > /tmp/r6rc/s_old_drain.hip
mk old_scalar_vmcnt0 /tmp/r6rc/s_old_drain.hip &
# 8 wait states in front of every asm load: cures it <=> VALU-writes-SGPR -> VMEM-reads-SGPR inside the asm statements
sed "$force; s/const int xr = w_r0 - 1 + w_j;  \/\/ x row delivered by this iteration/const int xr = w_r0 - 1 + w_j; asm volatile(\"s_nop 7\");/" $OLD > /tmp/r6rc/s_old_nop.hip
mk old_scalar_snop /tmp/r6rc/s_old_nop.hip &
wait
# read-write ("+v") destinations: the register holds a value the compiler must keep across the statement
sed "$force; s/: \"=v\"(sx\[set\])/: \"+v\"(sx[set])/; s/: \"=v\"(se\[set\])/: \"+v\"(se[set])/; s/: \"=v\"(sz\[set\])/: \"+v\"(sz[set])/" $OLD > /tmp/r6rc/s_old_rw.hip
mk old_scalar_rw /tmp/r6rc/s_old_rw.hip &
# the shipped round-5 dispatch (packed AFF build), control
mk old_packed $OLD &
# round-6 source: unconditional padded walk, scalar AFF forced / shipped dispatch
mk new_scalar dswgrad.hip -DDWG_SCALAR_AFF &
mk new_packed dswgrad.hip &
wait
ls -la ../exp
