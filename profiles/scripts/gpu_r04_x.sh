#!/bin/bash
# round 4, call X: core cycles per phase of the 8-phase K loop (ablation library, shader-clock stamps at the top of every phase of the third iteration of each tile).
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04_x; mkdir -p $O
export CVA_LIB=abl CVA_GEMM_DBG=294912 CVA_GEMM_PHASE=0
for s in "262144 5120 1280 0" "262144 1280 1280 0" "262144 1280 5120 0"; do timeout 300 python tools/experiments/r04_gemm_phases.py $s 2>&1 | grep -v amdgpu; done | tee $O/phases.txt
