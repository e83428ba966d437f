#!/bin/bash
# round 5, call m: attn2d_kernel<1> with the hazard-safe row sums and the softmax pieces re-balanced between the S^T and PV steps: tests (production library),
# the remaining ablation variants (prologue only, skeleton without DMA), the attention op and the bench step against the sequential kernel (ablation library,
# CVA_ATTN2D = 1 sequential / 2 in-wave pipeline / 3 pipeline with fragment prefetch distance 2).
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05_m; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_forward.py tests/test_gpu_fp8.py -x -q -m gpu -k "attention or samh or rows or race" > $O/tests.log 2>&1; echo "tests rc=$?" | tee -a $O/tests.log
tail -3 $O/tests.log
{
for d in 0 256 198 70 112; do
  echo -n "dbg $d: "; CVA_LIB=abl CVA_ATTN2D=2 CVA_ATTN2D_DBG=$d timeout 120 python tools/bench_attn.py 64 64 64 16 1280 0 10 2>&1 | grep -v amdgpu | tail -1 | sed 's/(qkv.*FLOPs/; FLOPs/'
done
for r in 1 2 3; do
  for v in 1 2 3; do
    echo -n "CVA_ATTN2D=$v: "; CVA_LIB=abl CVA_ATTN2D=$v timeout 120 python tools/bench_attn.py 64 64 64 16 1280 0 10 2>&1 | grep -v amdgpu | tail -1 | sed 's/(qkv.*FLOPs/; FLOPs/'
  done
done
} | tee $O/attn2d_ablation.txt
for r in 1 2; do
  for v in 1 2; do
    CVA_LIB=abl CVA_ATTN2D=$v timeout 300 python bench.py --allow-debug-env --no-extras --no-cpu-baseline --steps 6 > $O/bench_v${v}_$r.json 2> $O/bench_v${v}_$r.err
  done
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r05_m/bench_*.json')):
    try:
        r=json.load(open(f)); k=r['kernel_classes']
        print(f.split('/')[-1], round(r['value'],2), 'step', round(r['ms_per_step'],1), 'fwd', round(r['stage_ms_sequential']['forward'],1), 'attn', round(k['attention']['total_ms_per_step'],2), round(k['attention']['tflops']))
    except Exception as e: print(f, 'ERR', e)
PY
