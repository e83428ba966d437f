#!/bin/bash
# round 5, call j: attn2d_kernel with a ring of THREE stages (tile t+2 requested at the top of tile t, counted wait) against the two-stage form of call i:
# tests, the attention op on the production shape, the bench step; previous library against this one.
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05_j; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_gemm8.py tests/test_gpu_forward.py tests/test_gpu_fp8.py -x -q -m gpu -k "attention or samh or rows or race" > $O/tests.log 2>&1; echo "tests rc=$?" | tee -a $O/tests.log
tail -3 $O/tests.log
{
for r in 1 2 3; do
  echo -n "prev: "; CVA_LIB=libcellvit_amd_prev.so timeout 120 python tools/bench_attn.py 64 64 64 16 1280 0 10 2>&1 | grep -v amdgpu | tail -1
  echo -n "new : "; timeout 120 python tools/bench_attn.py 64 64 64 16 1280 0 10 2>&1 | grep -v amdgpu | tail -1
done
} | tee $O/attn_op_ab.txt
for r in 1 2; do
  CVA_LIB=libcellvit_amd_prev.so timeout 300 python bench.py --allow-debug-env --no-extras --no-cpu-baseline --steps 6 > $O/bench_prev_$r.json 2> $O/bench_prev_$r.err
  timeout 300 python bench.py --no-extras --no-cpu-baseline --steps 6 > $O/bench_new_$r.json 2> $O/bench_new_$r.err
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r05_j/bench_*.json')):
    try:
        r=json.load(open(f)); k=r['kernel_classes']
        print(f.split('/')[-1], round(r['value'],2), 'step', round(r['ms_per_step'],1), 'fwd', round(r['stage_ms_sequential']['forward'],1), 'attn', round(k['attention']['total_ms_per_step'],2), round(k['attention']['tflops']))
    except Exception as e: print(f, 'ERR', e)
PY
