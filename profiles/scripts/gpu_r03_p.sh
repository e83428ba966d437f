#!/bin/bash
OUT=gpurun_out/r03p; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_ops.py -q -x -k "attention" > $OUT/pytest_ops.log 2>&1; echo "ops rc=$?" > $OUT/rc.txt; tail -5 $OUT/pytest_ops.log
export CVA_LIB=abl
for rep in 1 2; do
CVA_ATTN=4 timeout 300 python tools/bench_attn.py >> $OUT/attn.txt 2>&1
for d in 0 1 2 3 4 8; do
  CVA_ATTN3_DBG=$d timeout 300 python tools/bench_attn.py >> $OUT/attn.txt 2>&1
done
done
grep -v amdgpu.ids $OUT/attn.txt | sed 's/ (qkv projection.*checksum/ cs/'
cat $OUT/rc.txt
