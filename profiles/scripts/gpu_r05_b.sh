#!/bin/bash
# round 5, call b: global attention with the softmax arithmetic inside the contraction (pre-scaled Q, accumulators initialised with the
# kw bias, the per-tile shift in the spare contraction slots of hd 80): op / forward / fp8 tests, then the previous library against this
# one on the bench step (alternating, same box, same call).
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05_b; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_forward.py tests/test_gpu_fp8.py -x -q -m gpu -k "attention or forward or rows" > $O/tests.log 2>&1; echo "tests rc=$?" | tee -a $O/tests.log
tail -3 $O/tests.log
for r in 1 2; do
  CVA_LIB=libcellvit_amd_prev.so timeout 300 python bench.py --allow-debug-env --no-extras --no-cpu-baseline --steps 6 > $O/bench_prev_$r.json 2> $O/bench_prev_$r.err
  timeout 300 python bench.py --no-extras --no-cpu-baseline --steps 6 > $O/bench_new_$r.json 2> $O/bench_new_$r.err
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r05_b/bench_*.json')):
    try:
        r=json.load(open(f)); k=r['kernel_classes']
        print(f.split('/')[-1], round(r['value'],2), 'fwd', round(r['stage_ms_sequential']['forward'],1), 'attn', round(k['attention']['total_ms_per_step'],2), 'lin', round(k['gemm_linear(proj/fc1/fc2/patch/neck)']['total_ms_per_step'],1))
    except Exception as e: print(f, 'ERR', e)
PY
