#!/bin/bash
# round 4, call L: what is the epilogue's cost made of?  fc1 / proj shapes at M = 262144, ablation flavour: CVA_GEMM_DBG 0 full; 1024 = everything computed, 25 % of the stores issued;
# 512 = half of the store instructions; 4 = no epilogue at all; 16 = next tile's DMA issued AFTER the epilogue instead of before.
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04_l; mkdir -p $O
export CVA_LIB=abl RACE=0
run() { timeout 300 python tools/bench_gemm.py $1 $2 $3 10 2>&1 | grep -v amdgpu.ids | grep -v RACE | tail -1; }
{
for shape in "262144 5120 1280 1 0" "262144 5120 1280 0 0" "262144 1280 1280 0 0"; do
  set -- $shape
  export ACT=$4 RES=$5
  echo "== $shape"
  for d in 0 1024 512 4 16 0; do echo -n "dbg $d: "; CVA_GEMM_PHASE=0 CVA_GEMM_DBG=$d run $1 $2 $3; done
done
} > $O/gemm8_epilogue_ablation.txt 2>&1
cat $O/gemm8_epilogue_ablation.txt
