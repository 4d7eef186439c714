#!/bin/bash
# sustained MFMA rate, clock and power for constant / random / changing operands (no memory traffic)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
hipcc -O3 --offload-arch=gfx950 scripts/probes/mfma_power_probe.hip -o /tmp/mfma_power_probe || exit 1
for mode in 0 1 2; do
  /tmp/mfma_power_probe $mode 5 &
  pid=$!
  sleep 2
  for i in 1 2 3; do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power" | tr '\n' ' '; echo; sleep 0.7; done
  wait $pid
done
