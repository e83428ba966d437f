#!/bin/bash
OUT=gpurun_out/r03t; mkdir -p $OUT
for rep in 1 2 3; do
  timeout 300 python tools/bench_attn.py >> $OUT/attn.txt 2>&1                       # production library: attention2 with hipcc's SLP packing
  CVA_LIB=abl CVA_ATTN=4 timeout 300 python tools/bench_attn.py >> $OUT/attn.txt 2>&1   # ablation flavour built with -fno-slp-vectorize on attention2/3
  CVA_LIB=abl CVA_ATTN=5 timeout 300 python tools/bench_attn.py >> $OUT/attn.txt 2>&1   # attention3 (explicit f32x4 code: still packed)
done
grep -v amdgpu.ids $OUT/attn.txt | sed 's/ (qkv projection.*checksum/ cs/'
