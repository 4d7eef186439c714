#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/${1:-r6l}
mkdir -p "$OUT"
for cfg in "2000 256 0" "2000 256 4" "5000 256 4" "5000 512 8"; do
  echo "== iters blocks: $cfg" >> "$OUT/window_probe.txt"
  timeout 300 ./smaat_unet_amd/exp/r6_window_probe $cfg >> "$OUT/window_probe.txt" 2>&1
done
cat "$OUT/window_probe.txt"
