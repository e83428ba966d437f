#!/bin/bash
# round 4, call K: additive ablation of the 8-phase linear GEMM at the production shapes of a 64-tile step (work-skipping instantiations, wrong results, timing only):
# CVA_GEMM_DBG 0 = full kernel, 1 = no operand DMA inside the K loop, 2 = no fragment reads, 3 = neither, 4 = no epilogue, 7 = MFMAs + barriers only.
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04_k; mkdir -p $O
export CVA_LIB=abl RACE=0
run() { timeout 300 python tools/bench_gemm.py $1 $2 $3 10 2>&1 | grep -v amdgpu.ids | grep -v RACE | tail -1; }
{
for shape in "262144 5120 1280 1 0" "262144 1280 5120 0 1" "262144 1280 1280 0 0"; do
  set -- $shape
  export ACT=$4 RES=$5
  echo "== $shape"
  for d in 0 1 2 3 4 7 0; do echo -n "dbg $d: "; CVA_GEMM_PHASE=0 CVA_GEMM_DBG=$d run $1 $2 $3; done
done
} > $O/gemm8_ablation.txt 2>&1
cat $O/gemm8_ablation.txt
