#!/bin/bash
# round 3, call E: gemm2 (two workgroups per CU) correctness + A/B
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r03_e; mkdir -p $O
export CVA_LIB=abl
for shape in "131072 5120 1280 1 0" "131072 1280 5120 0 1" "131072 1280 1280 0 0" "16384 3840 1280 0 0" "4096 4096 256 1 1"; do
  set -- $shape
  for v in 0 1; do
    ACT=$4 RES=$5 RACE=2 CVA_GEMM2=$v timeout 300 python tools/bench_gemm.py $1 $2 $3 10 2>&1 | grep -v amdgpu.ids | tail -2 | sed "s/^/GEMM2=$v /"
  done
done > $O/bench_gemm2.txt 2>&1
cat $O/bench_gemm2.txt
