#!/bin/bash
# pointwise-GEMM family: parity of every implementation variant + per-layer timing
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/${1:-pw}
mkdir -p "$OUT"
for impl in ${IMPLS:-1 2}; do
  for grp in dsconv_fwd pointwise_fwd; do
    SMAAT_PW_IMPL=$impl timeout 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "$grp" --tb=short -p no:cacheprovider > "$OUT/k_${grp}_impl$impl.log" 2>&1
    echo "impl$impl $grp exit=$? $(tail -1 "$OUT/k_${grp}_impl$impl.log")"
  done
done
for impl in ${BENCH_IMPLS:-0 1 2}; do
  SMAAT_PW_IMPL=$impl timeout 300 python scripts/layer_bench.py > "$OUT/layer_bench_impl$impl.txt" 2>&1
  echo "impl$impl layer_bench exit=$?"; tail -1 "$OUT/layer_bench_impl$impl.txt"
done
