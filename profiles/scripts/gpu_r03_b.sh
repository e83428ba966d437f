#!/bin/bash
# round 3, call B: gemm4 (K-tile stream) correctness + A/B + additive ablation
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r03_b; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_gemm8.py tests/test_gpu_ops.py -x -q > $O/pytest_gemm.txt 2>&1
tail -3 $O/pytest_gemm.txt
export CVA_LIB=abl
for shape in "131072 5120 1280 1 0" "131072 1280 5120 0 1" "131072 1280 1280 0 0"; do
  set -- $shape
  for v in 0 11 10 12 13 31 32 33 34 37; do
    ACT=$4 RES=$5 RACE=1 CVA_GEMM4=$v timeout 300 python tools/bench_gemm.py $1 $2 $3 10 2>&1 | grep -v amdgpu.ids | tail -2
  done
done > $O/bench_gemm.txt 2>&1
cat $O/bench_gemm.txt
