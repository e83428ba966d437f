#!/bin/bash
# Tuning experiments on the GPU box (ablation / build-flag variants of the library; nothing here is product code).
OUT=gpurun_out/${1:-exp}
mkdir -p $OUT
echo "== flood LDS footprint (POOL_LDS, BITMAP_WORDS): post-processing ms per 32 tiles" > $OUT/flood.txt
for cfg in "1024 2048" "256 1024" "256 2048" "512 1024" "128 1024"; do
  set -- $cfg
  CVA_BUILD_FLAGS="-DCVA_POOL_LDS=$1 -DCVA_BITMAP_WORDS=$2" python -m cellvit_amd.build > /dev/null 2>&1
  echo "POOL_LDS=$1 BITMAP_WORDS=$2: $(python tools/bench_pp.py 32 5 2>&1 | tail -1)" >> $OUT/flood.txt
  if [ "$1" != "1024" ]; then CVA_BUILD_FLAGS="-DCVA_POOL_LDS=$1 -DCVA_BITMAP_WORDS=$2" python -m pytest tests/test_gpu_postproc.py -m gpu -q -x 2>&1 | tail -1 >> $OUT/flood.txt; fi
done
cat $OUT/flood.txt
echo "== conv ablation (CVA_CONV_DBG: 1 no weight DMA after chunk 0, 2 no halo DMA, 3 neither, 4 no MFMA)" > $OUT/conv_abl.txt
python -m cellvit_amd.build --ablation > /dev/null 2>&1
for dbg in 0 1 2 3 4 7; do
  echo "-- CVA_CONV_DBG=$dbg" >> $OUT/conv_abl.txt
  CVA_CONV_DBG=$dbg CONV_SHAPES=0,1,3,5,6,7 python tools/bench_conv.py 10 2>&1 | grep -v amdgpu.ids >> $OUT/conv_abl.txt
done
cat $OUT/conv_abl.txt
python -m cellvit_amd.build > /dev/null 2>&1
