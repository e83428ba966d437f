#!/bin/bash
# attention2 prologue: table blocks outside the queries' band skipped — A/B against the previous object
OUT=gpurun_out/r03y2; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_ops.py -q -x -k "attention" > $OUT/pytest_ops.log 2>&1; echo "ops rc=$?" > $OUT/rc.txt; tail -2 $OUT/pytest_ops.log
for rep in 1 2; do
for lib in libcellvit_amd_olda2.so abl; do
  CVA_LIB=$lib timeout 600 python bench.py --allow-debug-env --no-cpu-baseline --no-extras --no-postproc --steps 3 --warmup 1 > $OUT/b_$lib.json 2>> $OUT/err.txt
  python - <<PY
import json
d=json.loads(open("$OUT/b_$lib.json").read().strip().splitlines()[-1])
kc=d["kernel_classes"]
print("$lib", round(d["value"],2), round(d["ms_per_step"],1), {k.split("(")[0]:(round(v["total_ms_per_step"],2), round(v["tflops"])) for k,v in kc.items() if "att" in k})
PY
done
done
timeout 1200 python -m pytest tests/test_gpu_forward.py -q -x -k "samh" > $OUT/pytest_fwd.log 2>&1; echo "fwd rc=$?" >> $OUT/rc.txt; tail -2 $OUT/pytest_fwd.log
cat $OUT/rc.txt
