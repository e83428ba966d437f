#!/bin/bash
# round 6, call c: deconv.hip (ConvT o conv3x3 composed, halo kernel for the Cout 128 / 64 stages) — parity, then same-call A/B on the ablation library
# (CVA_DECONV_HALO4 = 0: two-launch form, 1: the new kernel)
O=gpurun_out/r06_c; mkdir -p $O
python -m pytest tests/test_gpu_ops.py -q -k "deconv_block" > $O/pytest_deconv.log 2>&1; tail -3 $O/pytest_deconv.log
python -m pytest tests/test_gpu_forward.py tests/test_gpu_product_route.py -q > $O/pytest_forward.log 2>&1; tail -3 $O/pytest_forward.log
for v in 0 1 0 1; do
  CVA_LIB=abl CVA_DECONV_HALO4=$v timeout 600 python bench.py --no-cpu-baseline --no-extras --allow-debug-env --steps 6 > $O/b.log 2>$O/b.err
  python - "$v" $O/b.log <<'PY' | tee -a $O/bench_ab.txt
import json, sys
r = json.loads([l for l in open(sys.argv[2]) if l.startswith("{")][-1])
print("CVA_DECONV_HALO4=%s %.2f tiles/s %.1f ms fwd %.1f" % (sys.argv[1], r["value"], r["ms_per_step"], r["stage_ms_sequential"]["forward"]),
      {k.split("(")[0]: (v["launches"], round(v["total_ms_per_step"], 1), round(v["tflops"])) for k, v in r["kernel_classes"].items()})
PY
done
for v in 0 1; do
  CVA_LIB=abl CVA_DECONV_HALO4=$v timeout 600 python bench.py --model vit256 --no-cpu-baseline --no-extras --allow-debug-env --steps 6 > $O/b.log 2>$O/b.err
  python - "$v" $O/b.log <<'PY' | tee -a $O/bench_ab.txt
import json, sys
r = json.loads([l for l in open(sys.argv[2]) if l.startswith("{")][-1])
print("vit256 CVA_DECONV_HALO4=%s %.2f tiles/s %.1f ms" % (sys.argv[1], r["value"], r["ms_per_step"]),
      {k.split("(")[0]: (v["launches"], round(v["total_ms_per_step"], 1), round(v["tflops"])) for k, v in r["kernel_classes"].items()})
PY
done
