#!/bin/bash
# CellViT-256: the 312-channel bottleneck stage stored as 320 channels (halo kernels instead of the generic gather kernel)
OUT=gpurun_out/r03pad; mkdir -p $OUT
timeout 1500 python -m pytest tests/test_gpu_forward.py -q -x -k "vit256 or autocast or errors" > $OUT/pytest_fwd.log 2>&1; echo "fwd rc=$?" > $OUT/rc.txt; tail -4 $OUT/pytest_fwd.log
for rep in 1 2; do
  timeout 600 python bench.py --model vit256 --no-cpu-baseline --no-extras > $OUT/b_vit256_$rep.json 2>> $OUT/err.txt
  python - <<PY
import json
d=json.loads(open("$OUT/b_vit256_$rep.json").read().strip().splitlines()[-1])
kc=d["kernel_classes"]
print("vit256", round(d["value"],2), round(d["ms_per_step"],1), {k.split("(")[0]:(round(v["total_ms_per_step"],2), round(v["tflops"])) for k,v in kc.items()})
PY
done
timeout 1500 python -m pytest tests/test_gpu_forward.py tests/test_gpu_product_route.py tests/test_cli.py -q -x -m gpu > $OUT/pytest_all.log 2>&1; echo "all rc=$?" >> $OUT/rc.txt; tail -3 $OUT/pytest_all.log
cat $OUT/rc.txt
