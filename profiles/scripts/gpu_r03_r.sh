#!/bin/bash
# CellViT-256: linear layers on the 8-phase kernel with padded extents — goldens + A/B (CVA_GEMM_PAD=0/1, ablation flavour)
OUT=gpurun_out/r03r; mkdir -p $OUT
timeout 1500 python -m pytest tests/test_gpu_forward.py -q -x -k "vit256" -s > $OUT/pytest_fwd.log 2>&1; echo "fwd rc=$?" > $OUT/rc.txt; grep -E "batch 8|passed|failed|Error|assert" $OUT/pytest_fwd.log | tail -12
for v in 0 1 0 1; do
  CVA_LIB=abl CVA_GEMM_PAD=$v timeout 600 python bench.py --model vit256 --allow-debug-env --no-cpu-baseline --no-extras --steps 4 --warmup 2 > $OUT/bench_abl_pad$v.json 2>> $OUT/bench_abl.err
  python - <<PY
import json
d=json.loads(open("$OUT/bench_abl_pad$v.json").read().strip().splitlines()[-1])
kc=d["kernel_classes"]
print("pad=$v", round(d["value"],2), round(d["ms_per_step"],1), {k.split("(")[0]:(round(v["total_ms_per_step"],1), round(v["tflops"])) for k,v in kc.items()})
PY
done
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_fp8.py -q -x > $OUT/pytest_ops.log 2>&1; echo "ops rc=$?" >> $OUT/rc.txt; tail -3 $OUT/pytest_ops.log
cat $OUT/rc.txt
