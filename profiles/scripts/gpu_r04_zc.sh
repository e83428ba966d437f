#!/bin/bash
# round 4, call ZC: balanced DMA schedule also for the implicit-GEMM convolutions / composed launches (per-call set-up recomputed, LDS addresses as immediates):
# op tests, bit-identity of the outputs with the previous commit's library at production shapes, then the layer shapes, alternating.
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04_zc; mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_gemm8.py tests/test_gpu_ops.py -x -q -m gpu > $O/tests.log 2>&1; echo "tests rc=$?" | tee -a $O/tests.log
tail -2 $O/tests.log
CVA_LIB=libcellvit_amd_prev.so timeout 200 python tools/experiments/r04_conv_identity.py 2>&1 | grep -v amdgpu > $O/id_prev.txt
timeout 200 python tools/experiments/r04_conv_identity.py 2>&1 | grep -v amdgpu > $O/id_new.txt
cat $O/id_new.txt; diff $O/id_prev.txt $O/id_new.txt && echo "IDENTICAL to the previous library" | tee $O/identity.txt
{
for r in 1 2; do
  echo "prev:"; CVA_LIB=libcellvit_amd_prev.so CONV_SHAPES=0,1,2,3 timeout 100 python tools/bench_conv.py 10 2>&1 | grep -v amdgpu.ids
  echo "new :"; CVA_LIB= CONV_SHAPES=0,1,2,3 timeout 100 python tools/bench_conv.py 10 2>&1 | grep -v amdgpu.ids
done
} > $O/conv_ab.txt 2>&1
cat $O/conv_ab.txt
