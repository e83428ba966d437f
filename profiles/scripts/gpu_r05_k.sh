#!/bin/bash
# round 5, call k: attn2d_kernel<1> — S^T of key tile t + 1 issued beside the softmax of tile t inside the wave (fenced issue order) — against attn2d_kernel<0>:
# tests with the production library (which dispatches <1>), then the ablation library with CVA_ATTN2D = 1 / 2 alternating: the attention op on the production
# shape (checksums must be equal: the arithmetic per score is the same) and the bench step.
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05_k; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_gemm8.py tests/test_gpu_forward.py tests/test_gpu_fp8.py -x -q -m gpu -k "attention or samh or rows or race" > $O/tests.log 2>&1; echo "tests rc=$?" | tee -a $O/tests.log
tail -3 $O/tests.log
{
for r in 1 2 3; do
  echo -n "seq : "; CVA_LIB=abl CVA_ATTN2D=1 timeout 120 python tools/bench_attn.py 64 64 64 16 1280 0 10 2>&1 | grep -v amdgpu | tail -1
  echo -n "pipe: "; CVA_LIB=abl CVA_ATTN2D=2 timeout 120 python tools/bench_attn.py 64 64 64 16 1280 0 10 2>&1 | grep -v amdgpu | tail -1
done
} | tee $O/attn_op_ab.txt
for r in 1 2; do
  CVA_LIB=abl CVA_ATTN2D=1 timeout 300 python bench.py --allow-debug-env --no-extras --no-cpu-baseline --steps 6 > $O/bench_seq_$r.json 2> $O/bench_seq_$r.err
  CVA_LIB=abl CVA_ATTN2D=2 timeout 300 python bench.py --allow-debug-env --no-extras --no-cpu-baseline --steps 6 > $O/bench_pipe_$r.json 2> $O/bench_pipe_$r.err
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r05_k/bench_*.json')):
    try:
        r=json.load(open(f)); k=r['kernel_classes']
        print(f.split('/')[-1], round(r['value'],2), 'step', round(r['ms_per_step'],1), 'fwd', round(r['stage_ms_sequential']['forward'],1), 'attn', round(k['attention']['total_ms_per_step'],2), round(k['attention']['tflops']))
    except Exception as e: print(f, 'ERR', e)
PY
