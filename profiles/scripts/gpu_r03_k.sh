#!/bin/bash
# A/B: counted waits relaxed past the previous tile's epilogue stores (CVA_GEMM_DBG=8192, ablation build)
OUT=gpurun_out/r03k; mkdir -p $OUT
export CVA_LIB=abl
for rep in 1 2; do
for shape in "131072 5120 1280" "131072 1280 5120" "131072 1280 1280" "131072 3840 1280"; do
  for dbg in 0 8192; do
    echo "== $shape dbg=$dbg" >> $OUT/gemm.txt
    ACT=$([ "$shape" == "131072 5120 1280" ] && echo 1 || echo 0) CVA_GEMM_DBG=$dbg timeout 300 python tools/bench_gemm.py $shape 20 >> $OUT/gemm.txt 2>&1
  done
done
done
cat $OUT/gemm.txt
