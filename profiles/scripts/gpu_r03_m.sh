#!/bin/bash
# branch stages (ConvTranspose2d + concat conv) as composed launches: op test, forward goldens, same-box A/B, production bench
OUT=gpurun_out/r03m; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_ops.py -q -x -k "deconv" > $OUT/pytest_ops.log 2>&1; echo "ops rc=$?" > $OUT/rc.txt; tail -15 $OUT/pytest_ops.log
timeout 1500 python -m pytest tests/test_gpu_forward.py -q -x > $OUT/pytest_fwd.log 2>&1; echo "fwd rc=$?" >> $OUT/rc.txt; tail -5 $OUT/pytest_fwd.log
for v in 0 1 3 0 1 3; do
  CVA_LIB=abl CVA_DECONV_COMP=$v timeout 600 python bench.py --allow-debug-env --no-cpu-baseline --no-extras --steps 4 --warmup 2 > $OUT/bench_abl_comp$v.json 2>> $OUT/bench_abl.err
  python - <<PY
import json
d=json.loads(open("$OUT/bench_abl_comp$v.json").read().strip().splitlines()[-1])
kc=d["kernel_classes"]
print("comp=$v", round(d["value"],2), round(d["ms_per_step"],1), {k.split("(")[0]:(round(v["total_ms_per_step"],1), round(v["tflops"])) for k,v in kc.items() if "conv" in k})
PY
done
timeout 600 python bench.py --no-cpu-baseline --no-extras > $OUT/bench_prod.json 2> $OUT/bench_prod.err; echo "bench rc=$?" >> $OUT/rc.txt
cat $OUT/rc.txt; python -c "
import json; d=json.loads(open('$OUT/bench_prod.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"
