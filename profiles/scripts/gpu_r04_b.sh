#!/bin/bash
# round 4, call B: phase-shifted tile walk of the 8-phase GEMM (self-parked first tile): correctness (gemm + op + forward tests),
# same-call A/B on the three linear shapes of a 64-tile step (CVA_GEMM_PHASE 0 / 1, ablation flavour), then the default bench line;
# plus the new CPU-side parity tests that need a GPU (regression_loss goldens, ring repair on the device, > 8 classes).
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04_b; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_gemm8.py tests/test_gpu_ops.py tests/test_stitch_rings.py tests/test_gpu_stitch.py -x -q -m gpu > $O/pytest_a.txt 2>&1; tail -3 $O/pytest_a.txt
timeout 1500 python -m pytest tests/test_gpu_forward.py tests/test_evaluate.py -x -q -m gpu > $O/pytest_b.txt 2>&1; tail -3 $O/pytest_b.txt
export CVA_LIB=abl
run() { timeout 300 python tools/bench_gemm.py $1 $2 $3 10 2>&1 | grep -v amdgpu.ids | tail -2; }
{
for shape in "262144 5120 1280 1 0" "262144 1280 5120 0 1" "262144 1280 1280 0 0" "131072 1280 5120 0 1" "131072 1280 1280 0 0"; do
  set -- $shape
  export ACT=$4 RES=$5 RACE=2
  echo "== $shape"
  for ph in 0 1 0 1; do echo "phase $ph"; CVA_GEMM_PHASE=$ph run $1 $2 $3; done
done
} > $O/bench_gemm_phase.txt 2>&1
cat $O/bench_gemm_phase.txt
unset CVA_LIB ACT RES RACE
python bench.py --no-cpu-baseline --no-extras > $O/bench_f16.json 2> $O/bench_f16.err; python - <<'PY'
import json
r=json.loads(open('gpurun_out/r04_b/bench_f16.json').read().strip().splitlines()[-1])
print(r['value'], r['ms_per_step'], r['stage_ms_sequential'])
for k,v in r['kernel_classes'].items(): print(k, round(v['tflops']), round(v['total_ms_per_step'],1))
PY
