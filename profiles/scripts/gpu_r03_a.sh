#!/bin/bash
# round 3, call A: gemm4 (one wave per SIMD) correctness + same-call A/B against the 8-wave kernel
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r03_a; mkdir -p $O
export CVA_LIB=abl
for shape in "131072 1280 1280 0 0" "131072 5120 1280 1 0" "131072 1280 5120 0 1"; do
  set -- $shape
  for v in 0 11 10 12 13; do
    ACT=$4 RES=$5 RACE=2 CVA_GEMM4=$v timeout 300 python tools/bench_gemm.py $1 $2 $3 20 2>&1 | tail -2
  done
done > $O/bench_gemm.txt 2>&1
unset CVA_LIB
timeout 900 python -m pytest tests/test_gpu_gemm8.py tests/test_gpu_ops.py -x -q > $O/pytest_gemm.txt 2>&1
timeout 600 python bench.py --steps 5 --warmup 2 > $O/bench.json 2> $O/bench.err
tail -3 $O/pytest_gemm.txt; cat $O/bench_gemm.txt; tail -c 1500 $O/bench.json
