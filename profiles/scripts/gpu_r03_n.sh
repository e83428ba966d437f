#!/bin/bash
# attention3 (8-wave counter-phase global attention): op tests, forward goldens, same-box A/B (CVA_ATTN=4 -> attention2)
OUT=gpurun_out/r03n; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_ops.py -q -x -k "attention" > $OUT/pytest_ops.log 2>&1; echo "ops rc=$?" > $OUT/rc.txt; tail -15 $OUT/pytest_ops.log
for v in 4 3 4 3; do
  CVA_LIB=abl CVA_ATTN=$v timeout 600 python bench.py --allow-debug-env --no-cpu-baseline --no-extras --steps 4 --warmup 2 > $OUT/bench_abl_attn$v.json 2>> $OUT/bench_abl.err
  python - <<PY
import json
d=json.loads(open("$OUT/bench_abl_attn$v.json").read().strip().splitlines()[-1])
kc=d["kernel_classes"]
print("attn=$v", round(d["value"],2), round(d["ms_per_step"],1), {k.split("(")[0]:(round(v["total_ms_per_step"],1), round(v["tflops"])) for k,v in kc.items() if "att" in k})
PY
done
timeout 1500 python -m pytest tests/test_gpu_forward.py -q -x > $OUT/pytest_fwd.log 2>&1; echo "fwd rc=$?" >> $OUT/rc.txt; tail -5 $OUT/pytest_fwd.log
cat $OUT/rc.txt
