#!/bin/bash
# round 4, call Z2: DMA pieces per phase 2 / 2 / 3 / 1 instead of 2 / 2 / 2 / 2 (one W piece moved from the twelve-read phases 4 / 8 into the read-free phases 3 / 7).
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04_z2; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_gemm8.py tests/test_gpu_ops.py -x -q -m gpu > $O/tests.log 2>&1; echo "tests rc=$?" | tee -a $O/tests.log
tail -2 $O/tests.log
run() { timeout 300 python tools/bench_gemm.py $1 $2 $3 10 2>&1 | grep -v amdgpu.ids | grep -v RACE | tail -1; }
{
for shape in "262144 5120 1280 1 0" "262144 1280 5120 0 1" "262144 1280 1280 0 0"; do
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
CVA_LIB=abl CVA_GEMM_DBG=294912 CVA_GEMM_PHASE=0 timeout 300 python tools/experiments/r04_gemm_phases.py 262144 1280 5120 0 2>&1 | grep -v amdgpu | tee $O/phases.txt
for r in 1 2; do
  for l in libcellvit_amd_prev.so ""; do
    echo "bench lib='$l'" >> $O/bench_ab.txt
    CVA_LIB=$l timeout 600 python bench.py --no-cpu-baseline --no-extras --allow-debug-env 2>/dev/null | grep '^{' >> $O/bench_ab.txt
  done
done
python - <<'PY'
import json
for ln in open("gpurun_out/r04_z2/bench_ab.txt"):
    if ln.startswith("{"):
        d = json.loads(ln); kc = d["kernel_classes"]
        print("   ", round(d["value"], 2), round(d["ms_per_step"], 1), {k.split("(")[0]: (round(v["total_ms_per_step"], 1), round(v["tflops"])) for k, v in kc.items()})
    else: print(ln.strip())
PY
