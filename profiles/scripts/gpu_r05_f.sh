#!/bin/bash
# round 5, call f: dynamic tile walk of the 8-phase kernel (per-XCD position counters, one returning atomic per tile issued in the K loop):
# GEMM / conv / forward / fp8 tests; the isolated linear shapes previous library against this one (the atomic's cost with nothing else resident);
# the bench step, where the post-processing of the previous batch is resident on a second stream (the walk's purpose), alternating.
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05_f; mkdir -p $O
timeout 1000 python -m pytest tests/test_gpu_gemm8.py tests/test_gpu_ops.py tests/test_gpu_forward.py tests/test_gpu_fp8.py -x -q -m gpu -k "not instance_level" > $O/tests.log 2>&1; echo "tests rc=$?" | tee -a $O/tests.log
tail -3 $O/tests.log
{
for shape in "262144 5120 1280" "262144 1280 1280"; do
  for r in 1 2; do
    echo -n "prev $shape: "; CVA_LIB=libcellvit_amd_prev.so ACT=1 timeout 100 python tools/bench_gemm.py $shape 10 2>&1 | grep -v amdgpu | tail -1
    echo -n "new  $shape: "; ACT=1 timeout 100 python tools/bench_gemm.py $shape 10 2>&1 | grep -v amdgpu | tail -1
  done
done
for r in 1 2; do
  echo -n "prev fc2: "; CVA_LIB=libcellvit_amd_prev.so RES=1 timeout 100 python tools/bench_gemm.py 262144 1280 5120 10 2>&1 | grep -v amdgpu | tail -1
  echo -n "new  fc2: "; RES=1 timeout 100 python tools/bench_gemm.py 262144 1280 5120 10 2>&1 | grep -v amdgpu | tail -1
done
} | tee $O/gemm_ab.txt
for r in 1 2 3; do
  CVA_LIB=libcellvit_amd_prev.so timeout 300 python bench.py --allow-debug-env --no-extras --no-cpu-baseline --steps 6 > $O/bench_prev_$r.json 2> $O/bench_prev_$r.err
  timeout 300 python bench.py --no-extras --no-cpu-baseline --steps 6 > $O/bench_new_$r.json 2> $O/bench_new_$r.err
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r05_f/bench_*.json')):
    try:
        r=json.load(open(f)); k=r['kernel_classes']
        print(f.split('/')[-1], round(r['value'],2), 'step', round(r['ms_per_step'],1), 'fwd', round(r['stage_ms_sequential']['forward'],1), 'pp', round(r['stage_ms_sequential']['postproc'],1), 'lin', round(k['gemm_linear(proj/fc1/fc2/patch/neck)']['total_ms_per_step'],1), 'qkv', round(k['gemm_qkv']['total_ms_per_step'],1), 'conv', round(k['conv3x3_implicit_gemm']['total_ms_per_step'],1))
    except Exception as e: print(f, 'ERR', e)
PY
