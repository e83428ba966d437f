#!/bin/bash
# round 4, call ZF: balanced DMA schedule in the fp8 K loop: fp8 op tests, then per shape the previous commit's library against this one — output digests (must be equal:
# same MFMAs in the same order) and timing, alternating.
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04_zf; mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_fp8.py -x -q -m gpu -k "not forward_fp8 and not instance_level" > $O/tests.log 2>&1; echo "tests rc=$?" | tee -a $O/tests.log
tail -2 $O/tests.log
{
for shape in "262144 5120 1280 10 2 1" "262144 1280 5120 10 1 0" "262144 1280 1536 10 0 0" "131072 3840 1280 10 0 0"; do
  for r in 1 2; do
    echo -n "prev: "; CVA_LIB=libcellvit_amd_prev.so RACE=1 timeout 100 python tools/bench_gemm_mx8.py $shape 2>&1 | grep -v amdgpu | tail -1
    echo -n "new : "; RACE=1 timeout 100 python tools/bench_gemm_mx8.py $shape 2>&1 | grep -v amdgpu | tail -1
  done
done
} | tee $O/mx8_ab.txt
