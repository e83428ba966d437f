#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r03_g; mkdir -p $O
export CVA_LIB=abl
for shape in "131072 5120 1280 1 0" "131072 1280 1280 0 0"; do
  set -- $shape
  for v in 0 11 14; do
    ACT=$4 RES=$5 RACE=2 CVA_GEMM4=$v timeout 300 python tools/bench_gemm.py $1 $2 $3 10 2>&1 | grep -v amdgpu.ids | tail -2
  done
done > $O/bench_gemm4_s4.txt 2>&1
cat $O/bench_gemm4_s4.txt
