#!/bin/bash
# round 5, call g: which change broke the repeated-launch race screen of the window attention op (call f)?  The same test under three libraries:
# oldwin (round-4 window kernel, static GEMM walk), prev (window rows, static walk), current (window rows, dynamic walk).
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05_g; mkdir -p $O
for lib in libcellvit_amd_oldwin.so libcellvit_amd_prev.so libcellvit_amd.so; do
  for r in 1 2; do
    echo "== $lib run $r" | tee -a $O/race.txt
    CVA_LIB=$lib timeout 200 python -m pytest tests/test_gpu_gemm8.py -x -q -m gpu -k "race_screen" 2>&1 | tail -2 | tee -a $O/race.txt
  done
done
