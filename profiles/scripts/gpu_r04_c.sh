#!/bin/bash
# round 4, call C: V row-major (qkv epilogue writes V like K; attn2_kernel / attnwp_kernel read it through ds_read_b64_tr_b16):
# the whole GPU suite on the production library, then a same-call A/B of the default bench step on the ablation flavour
# (CVA_NO_VRM=1 -> V^T as before).
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04_c; mkdir -p $O
timeout 2400 python -m pytest tests -x -q -m gpu > $O/pytest.txt 2>&1; tail -5 $O/pytest.txt
summ() { python - "$1" <<'PY'
import json, sys
r=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1], round(r['value'],2), 'tiles/s', round(r['ms_per_step'],1), 'ms', {k.split('(')[0]: (round(v['tflops']), round(v['total_ms_per_step'],1)) for k,v in r['kernel_classes'].items()})
PY
}
export CVA_LIB=abl
for rep in 1 2; do
  for nv in 1 0; do
    CVA_NO_VRM=$nv python bench.py --allow-debug-env --no-cpu-baseline --no-extras --steps 10 > $O/bench_novrm${nv}_$rep.json 2> $O/bench_novrm${nv}_$rep.err
    summ $O/bench_novrm${nv}_$rep.json
  done
done
unset CVA_LIB
python bench.py --no-cpu-baseline --no-extras > $O/bench_f16.json 2> $O/bench_f16.err; summ $O/bench_f16.json
python bench.py --no-cpu-baseline --no-extras --dtype f8 --steps 5 > $O/bench_f8.json 2> $O/bench_f8.err; summ $O/bench_f8.json
