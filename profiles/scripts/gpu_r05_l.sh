#!/bin/bash
# round 5, call l: what bounds attn2d_kernel<1>?  Ablation variants (CVA_ATTN2D_DBG, wrong results by construction) of the attention op on the production
# shape — the qkv projection in front of the kernel is the same ≈ 3.8 ms in every line —, then the fragment-prefetch distance 2 variant (CVA_ATTN2D=3) against
# distance 1 (=2) and the sequential kernel (=1), and the rebuilt production library's tests.
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05_l; mkdir -p $O
{
for d in 0 1 64 2 4 6 70 48 112 8 128 136; do
  echo -n "dbg $d: "; CVA_LIB=abl CVA_ATTN2D=2 CVA_ATTN2D_DBG=$d timeout 120 python tools/bench_attn.py 64 64 64 16 1280 0 10 2>&1 | grep -v amdgpu | tail -1 | sed 's/(qkv.*//'
done
for r in 1 2 3; do
  for v in 1 2 3; do
    echo -n "CVA_ATTN2D=$v: "; CVA_LIB=abl CVA_ATTN2D=$v timeout 120 python tools/bench_attn.py 64 64 64 16 1280 0 10 2>&1 | grep -v amdgpu | tail -1 | sed 's/(qkv.*//'
  done
done
} | tee $O/attn2d_ablation.txt
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_forward.py tests/test_gpu_fp8.py -x -q -m gpu -k "attention or samh or rows or race" > $O/tests.log 2>&1; echo "tests rc=$?" | tee -a $O/tests.log
tail -3 $O/tests.log
