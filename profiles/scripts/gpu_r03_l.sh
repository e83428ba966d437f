#!/bin/bash
# composed Deconv2DBlock: op test, forward goldens, same-box A/B (ablation flavour, CVA_DECONV_COMP=0/1), production bench
OUT=gpurun_out/r03l; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_ops.py -q -x -k "deconv or convT or implicit" > $OUT/pytest_ops.log 2>&1; echo "ops rc=$?" > $OUT/rc.txt; tail -5 $OUT/pytest_ops.log
timeout 1500 python -m pytest tests/test_gpu_forward.py -q -x > $OUT/pytest_fwd.log 2>&1; echo "fwd rc=$?" >> $OUT/rc.txt; tail -5 $OUT/pytest_fwd.log
for v in 0 1 0 1; do
  CVA_LIB=abl CVA_DECONV_COMP=$v timeout 600 python bench.py --allow-debug-env --no-cpu-baseline --no-extras --steps 4 --warmup 2 > $OUT/bench_abl_comp$v.json 2>> $OUT/bench_abl.err
  python - <<PY
import json
d=json.loads(open("$OUT/bench_abl_comp$v.json").read().strip().splitlines()[-1])
print("comp=$v", d["value"], d["ms_per_step"], {k:v for k,v in d.get("classes",{}).items()} if "classes" in d else "")
PY
done
timeout 600 python bench.py --no-cpu-baseline --no-extras > $OUT/bench_prod.json 2> $OUT/bench_prod.err; echo "bench rc=$?" >> $OUT/rc.txt
cat $OUT/rc.txt; cat $OUT/bench_prod.json
