#!/bin/bash
# Round 6: build libsmaat_hip with dswgrad.hip's DEVICE code taken from a hand-edited assembly listing (ISA-level experiments on
# the nondeterministic instantiation: the compiler is out of the loop, only the named edit differs between two libraries).
#   r6_asm_variant.sh <tag> <source.hip> <sed-script-or-python-editor> [extra hipcc flags]
# steps: device asm (-S) -> edit -> assemble -> lld -> clang-offload-bundler -> host object with that fat binary -> link
set -eu
tag=$1; src=$2; edit=$3; shift 3
CS="$(cd "$(dirname "$0")/../../smaat_unet_amd/csrc" && pwd)"
LL=/opt/rocm/lib/llvm/bin
W=/tmp/r6rc/asm_$tag; mkdir -p $W "$CS/../exp"
INC="-I$CS -I$CS/../../include"
cp "$src" $W/dswgrad.hip
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC $INC "$@" --cuda-device-only -S $W/dswgrad.hip -o $W/dev.s 2>/dev/null
python3 "$edit" $W/dev.s $W/dev_edit.s "$tag"
$LL/clang -x assembler -target amdgcn-amd-amdhsa -mcpu=gfx950 -c $W/dev_edit.s -o $W/dev.o
$LL/lld -flavor gnu -m elf64_amdgpu --no-undefined -shared $W/dev.o -o $W/dev.co
$LL/clang-offload-bundler -type=o -bundle-align=4096 -targets=host-x86_64-unknown-linux-gnu,hipv4-amdgcn-amd-amdhsa--gfx950 -input=/dev/null -input=$W/dev.co -output=$W/dev.hipfb
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC $INC "$@" --cuda-host-only -Xclang -fcuda-include-gpubinary -Xclang $W/dev.hipfb -c $W/dswgrad.hip -o $W/host.o 2>/dev/null
others=$(ls $CS/*.o | grep -v "/dswgrad.o$" | tr '\n' ' ')
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $others $W/host.o -o "$CS/../exp/libsmaat_hip_rc_asm_$tag.so"
echo "built asm_$tag"
