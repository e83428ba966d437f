#!/bin/bash
# round 4, call F: fp8 proj (attention kernels emit MX-fp8 rows, proj on the block-scaled MFMA) + sub-batch pipelined post-processing.
# (1) whole GPU suite; (2) post-processing alone and the bench step, CVA_PP_SPLIT=0 / 1 (ablation flavour, same call); (3) bench lines f16 / f8.
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04_f; mkdir -p $O
timeout 2700 python -m pytest tests -q -m gpu > $O/pytest.txt 2>&1; tail -6 $O/pytest.txt; grep -n "FAILED\|rows mx8\|proj mx8\|fp8\]" $O/pytest.txt | head -20
summ() { python - "$1" <<'PY'
import json, sys
r=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1], round(r['value'],2), 'tiles/s', round(r['ms_per_step'],1), 'ms', r['stage_ms_sequential'], {k.split('(')[0]: (round(v['tflops']), round(v['total_ms_per_step'],1)) for k,v in r['kernel_classes'].items()})
PY
}
export CVA_LIB=abl
for rep in 1 2; do
  for sp in 0 1; do
    CVA_PP_SPLIT=$sp python tools/bench_pp.py 64 5 2>&1 | tail -1 | sed "s/^/split $sp: /"
    CVA_PP_SPLIT=$sp python bench.py --allow-debug-env --no-cpu-baseline --no-extras --steps 10 > $O/bench_split${sp}_$rep.json 2> $O/bench_split${sp}_$rep.err; summ $O/bench_split${sp}_$rep.json
  done
done
unset CVA_LIB
python bench.py --no-cpu-baseline --no-extras > $O/bench_f16.json 2> $O/bench_f16.err; summ $O/bench_f16.json
python bench.py --no-cpu-baseline --no-extras --dtype f8 > $O/bench_f8.json 2> $O/bench_f8.err; summ $O/bench_f8.json
