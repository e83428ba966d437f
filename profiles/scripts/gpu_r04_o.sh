#!/bin/bash
# round 4, call O: gemm8 with the lean linear epilogue row loops (and without register spills) against the previous commit's library on the same box:
# correctness first, then the linear shapes of a 64-tile step, the per-tile timeline, and the end-to-end bench (alternating, two rounds).
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04_o; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_gemm8.py tests/test_gpu_ops.py tests/test_gpu_forward.py tests/test_gpu_fp8.py -x -q -m gpu > $O/tests.log 2>&1; echo "tests rc=$?" | tee -a $O/tests.log
tail -3 $O/tests.log
run() { timeout 300 python tools/bench_gemm.py $1 $2 $3 10 2>&1 | grep -v amdgpu.ids | grep -v RACE | tail -1; }
{
for shape in "262144 5120 1280 1 0" "262144 1280 5120 0 1" "262144 1280 1280 0 0" "262144 3840 1280 0 0"; do
  set -- $shape
  export ACT=$4 RES=$5
  echo "== $shape"
  for r in 1 2; do
    echo -n "prev: "; CVA_LIB=libcellvit_amd_prev.so run $1 $2 $3
    echo -n "new : "; CVA_LIB= run $1 $2 $3
  done
done
unset ACT RES
} > $O/gemm_ab.txt 2>&1
cat $O/gemm_ab.txt
{
export CVA_LIB=abl CVA_GEMM_DBG=32768 CVA_GEMM_PHASE=0
for s in "262144 5120 1280 1" "262144 5120 1280 0" "262144 1280 1280 0" "262144 1280 5120 0"; do python tools/experiments/r04_gemm_timeline.py $s 2>&1 | grep -v amdgpu; done
unset CVA_LIB CVA_GEMM_DBG CVA_GEMM_PHASE
} > $O/timeline.txt 2>&1
cat $O/timeline.txt
for r in 1 2; do
  for l in libcellvit_amd_prev.so ""; do
    echo "bench lib='$l'" >> $O/bench_ab.txt
    CVA_LIB=$l timeout 600 python bench.py --no-cpu-baseline --no-extras > $O/bench_last.log 2>&1; grep '^{' $O/bench_last.log >> $O/bench_ab.txt || tail -5 $O/bench_last.log
  done
done
python - <<'PY'
import json
for ln in open("gpurun_out/r04_o/bench_ab.txt"):
    if ln.startswith("{"):
        d = json.loads(ln); print("   ", d["value"], d["ms_per_step"], d["roofline"]["frac"])
    else: print(ln.strip())
PY
