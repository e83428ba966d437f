#!/bin/bash
# round 5, call q: the final tree's bench lines on a second box of the pool (the evidence call r05_zz landed on a slow one), and the fp8 line with its own parity bounds.
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05_q; mkdir -p $O
python bench.py --no-cpu-baseline --no-extras > $O/bench_f16.json 2> $O/bench_f16.err
python bench.py --dtype f8 --no-cpu-baseline --no-extras > $O/bench_f8.json 2> $O/bench_f8.err
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
python - <<'PY'
import json
for n in ('f16','f8'):
    d=json.loads(open(f'gpurun_out/r05_q/bench_{n}.json').read().strip().splitlines()[-1])
    print(n, round(d['value'],2), round(d['ms_per_step'],1), {k.split('(')[0]:(round(v['total_ms_per_step'],1),round(v['tflops'])) for k,v in d['kernel_classes'].items()}, 'parity', d['parity']['forward']['pass'], d['parity']['postproc']['pass'])
PY
