#!/bin/bash
# round 4, call P: direct (LDS-free) epilogue of the 4-wave halo convolution + lean linear epilogue rows of gemm8, against the previous commit's
# library on the same box: correctness first, then the decoder's layer shapes, the linear shapes, and the end-to-end bench (alternating, two rounds).
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04_p; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_ops.py tests/test_gpu_forward.py tests/test_gpu_gemm8.py tests/test_gpu_fp8.py -x -q -m gpu > $O/tests.log 2>&1; echo "tests rc=$?" | tee -a $O/tests.log
tail -3 $O/tests.log
{
for r in 1 2; do
  echo "prev:"; CVA_LIB=libcellvit_amd_prev.so CONV_SHAPES=4,5,6,7 timeout 300 python tools/bench_conv.py 10 2>&1 | grep -v amdgpu.ids
  echo "new :"; CVA_LIB= CONV_SHAPES=4,5,6,7 timeout 300 python tools/bench_conv.py 10 2>&1 | grep -v amdgpu.ids
done
} > $O/conv_ab.txt 2>&1
cat $O/conv_ab.txt
for r in 1 2; do
  for l in libcellvit_amd_prev.so ""; do
    echo "bench lib='$l'" >> $O/bench_ab.txt
    CVA_LIB=$l timeout 600 python bench.py --no-cpu-baseline --no-extras --allow-debug-env > $O/bench_last.log 2>&1; grep '^{' $O/bench_last.log >> $O/bench_ab.txt || tail -5 $O/bench_last.log
  done
done
python - <<'PY'
import json
for ln in open("gpurun_out/r04_p/bench_ab.txt"):
    if ln.startswith("{"):
        d = json.loads(ln); kc = d["kernel_classes"]
        print("   ", round(d["value"], 2), round(d["ms_per_step"], 1), {k.split("(")[0]: round(v["total_ms_per_step"], 1) for k, v in kc.items()})
    else: print(ln.strip())
PY
