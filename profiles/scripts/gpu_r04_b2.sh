#!/bin/bash
# round 4, call B2: the phase-shifted walk again (the first build copied / spilled 72 accumulator VGPRs around every epilogue: 36 % slower),
# now with the parked sums added to the epilogue's per-row temporaries; also the nearest-node GELU table.  Then call C (V row-major).
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04_b2; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_gemm8.py tests/test_gpu_ops.py -x -q -m gpu > $O/pytest_a.txt 2>&1; tail -3 $O/pytest_a.txt
export CVA_LIB=abl
run() { timeout 300 python tools/bench_gemm.py $1 $2 $3 10 2>&1 | grep -v amdgpu.ids | tail -2; }
{
for shape in "262144 5120 1280 1 0" "262144 1280 5120 0 1" "262144 1280 1280 0 0" "131072 1280 5120 0 1"; do
  set -- $shape
  export ACT=$4 RES=$5 RACE=2
  echo "== $shape"
  for ph in 0 1 0 1; do echo "phase $ph"; CVA_GEMM_PHASE=$ph run $1 $2 $3; done
done
} > $O/bench_gemm_phase.txt 2>&1
cat $O/bench_gemm_phase.txt
unset CVA_LIB ACT RES RACE
bash tools/experiments/gpu_r04_c.sh
