#!/bin/bash
OUT=gpurun_out/r03o; mkdir -p $OUT
export CVA_LIB=abl
for rep in 1 2; do
CVA_ATTN=4 timeout 300 python tools/bench_attn.py >> $OUT/attn.txt 2>&1
for d in 0 1 2 3 4 8 5 6 12; do
  CVA_ATTN3_DBG=$d timeout 300 python tools/bench_attn.py >> $OUT/attn.txt 2>&1
done
done
grep -v amdgpu.ids $OUT/attn.txt
