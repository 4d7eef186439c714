#!/bin/bash
# one gpurun call: parity tests + smoke + bench, per-layer bench, rocprofv3 evidence
cd "${GRAFT_REPO_ROOT:-/root/repo}"
TAG=${1:-r1}
bash scripts/gpu_round.sh $TAG
LB_OUT=gpurun_out/$TAG/layer_bench.json timeout 300 python scripts/layer_bench.py > gpurun_out/$TAG/layer_bench.txt 2>&1
echo "layer_bench exit=$?"; tail -3 gpurun_out/$TAG/layer_bench.txt
bash scripts/prof_round.sh $TAG/prof
