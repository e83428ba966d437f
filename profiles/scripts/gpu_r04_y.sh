#!/bin/bash
# round 4, call Y: balanced DMA schedule of the fp16 K loop (two pieces per wave and phase instead of four in phases 3, 4, 7, 8; waves stage two pieces of every
# sub-tile) against the previous commit's library on one box: correctness, the linear shapes, core cycles per phase, conv shapes, end-to-end bench.
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04_y; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_gemm8.py tests/test_gpu_ops.py tests/test_gpu_forward.py tests/test_gpu_fp8.py -x -q -m gpu > $O/tests.log 2>&1; echo "tests rc=$?" | tee -a $O/tests.log
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
export CVA_LIB=abl CVA_GEMM_DBG=294912 CVA_GEMM_PHASE=0
for s in "262144 5120 1280 0" "262144 1280 1280 0" "262144 1280 5120 0"; do timeout 300 python tools/experiments/r04_gemm_phases.py $s 2>&1 | grep -v amdgpu; done
unset CVA_LIB CVA_GEMM_DBG CVA_GEMM_PHASE
} > $O/phases.txt 2>&1
cat $O/phases.txt
{
for r in 1 2; do
  echo "prev:"; CVA_LIB=libcellvit_amd_prev.so CONV_SHAPES=0,1,2,3 timeout 300 python tools/bench_conv.py 10 2>&1 | grep -v amdgpu.ids
  echo "new :"; CVA_LIB= CONV_SHAPES=0,1,2,3 timeout 300 python tools/bench_conv.py 10 2>&1 | grep -v amdgpu.ids
done
} > $O/conv_ab.txt 2>&1
cat $O/conv_ab.txt
for r in 1 2; do
  for l in libcellvit_amd_prev.so ""; do
    echo "bench lib='$l'" >> $O/bench_ab.txt
    CVA_LIB=$l timeout 600 python bench.py --no-cpu-baseline --no-extras --allow-debug-env > $O/bench_last.log 2>&1; grep '^{' $O/bench_last.log >> $O/bench_ab.txt || tail -5 $O/bench_last.log
  done
done
python - <<'PY'
import json
for ln in open("gpurun_out/r04_y/bench_ab.txt"):
    if ln.startswith("{"):
        d = json.loads(ln); kc = d["kernel_classes"]
        print("   ", round(d["value"], 2), round(d["ms_per_step"], 1), {k.split("(")[0]: (round(v["total_ms_per_step"], 1), round(v["tflops"])) for k, v in kc.items()})
    else: print(ln.strip())
PY
