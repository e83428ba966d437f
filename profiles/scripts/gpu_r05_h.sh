#!/bin/bash
# round 5, call h: (1) ordered flood with a two-level minimum (values first, 128-bit keys only among exact ties) and no integer division on the pop chain:
# bit-exact post-processing tests, chain alone previous library against this one (64 and 8 tiles); (2) the CU-mask experiment VERDICT r04 asked to RUN:
# the post-processing stream of the bench step on a hipExtStreamCreateWithCUMask stream of 8 / 16 / 32 CUs (spread or lowest-numbered), forward unchanged.
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05_h; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_postproc.py tests/test_gpu_product_route.py -x -q -m gpu > $O/tests.log 2>&1; echo "tests rc=$?" | tee -a $O/tests.log
tail -3 $O/tests.log
{
for B in 64 8; do for r in 1 2; do
  echo -n "prev B=$B: "; CVA_LIB=libcellvit_amd_prev.so timeout 200 python tools/bench_pp.py $B 10 2>&1 | grep -v amdgpu | tail -1
  echo -n "new  B=$B: "; timeout 200 python tools/bench_pp.py $B 10 2>&1 | grep -v amdgpu | tail -1
done; done
} | tee $O/pp_ab.txt
for m in none 8 none 16 low8 32 none; do
  if [ $m == none ]; then a=""; else a="--pp-cu-mask $m"; fi
  timeout 300 python bench.py --no-extras --no-cpu-baseline --steps 6 $a > $O/bench_mask_$m.$RANDOM.json 2> $O/bench_mask_$m.err
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r05_h/bench_mask_*.json')):
    try:
        r=json.load(open(f))
        print(f.split('/')[-1], round(r['value'],2), 'step', round(r['ms_per_step'],1), 'fwd', round(r['stage_ms_sequential']['forward'],1), 'pp', round(r['stage_ms_sequential']['postproc'],1), r['config']['experiment_env'])
    except Exception as e: print(f, 'ERR', e)
PY
