#!/bin/bash
# sustained rate / clock / power of bf16 vs fp16 MFMA chains on random operands (no memory traffic): the pipe question behind
# the two-term fp16 split of DESIGN 4.7
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/mfma_f16
hipcc -O3 --offload-arch=gfx950 scripts/probes/mfma_power_probe.hip -o /tmp/p_bf16 2>/dev/null || exit 1
hipcc -O3 --offload-arch=gfx950 -DPROBE_F16=1 scripts/probes/mfma_power_probe.hip -o /tmp/p_f16 2>/dev/null || exit 1
for w in bf16 f16 bf16 f16; do
  echo "== $w, random operands (mode 2)"
  /tmp/p_$w 2 4 &
  pid=$!
  sleep 2
  for i in 1 2; do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power" | tr '\n' ' '; echo; sleep 0.6; done
  wait $pid
done 2>&1 | tee gpurun_out/mfma_f16/mfma_power_bf16_vs_f16.txt
