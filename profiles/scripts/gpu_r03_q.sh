#!/bin/bash
OUT=gpurun_out/r03q3; mkdir -p $OUT
export CVA_LIB=abl
for rep in 1 2 3; do
for d in 11 8 9 10 12 72 76 73 74; do
  CVA_ATTN3_DBG=$d timeout 300 python tools/bench_attn.py >> $OUT/attn.txt 2>&1
done
done
grep -v amdgpu.ids $OUT/attn.txt | sed 's/ (qkv projection.*checksum/ cs/' | awk '{print $2, $9}' | sort | awk '{a[$1]=a[$1]" "$2} END{for(k in a) print k, a[k]}'
