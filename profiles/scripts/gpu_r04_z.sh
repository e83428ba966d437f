#!/bin/bash
# round 4, call Z: balanced DMA schedule for the linear launches only (production library), additionally for the qkv projection (libcellvit_amd_q1.so), against the
# previous commit's library; convolutions keep the old schedule in both.  Correctness first, then the end-to-end bench, three libraries alternating, two rounds.
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04_z; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_gemm8.py tests/test_gpu_ops.py tests/test_gpu_forward.py tests/test_gpu_fp8.py -x -q -m gpu > $O/tests.log 2>&1; echo "tests rc=$?" | tee -a $O/tests.log
tail -3 $O/tests.log
CVA_LIB=libcellvit_amd_q1.so timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_forward.py -x -q -m gpu > $O/tests_q1.log 2>&1; echo "tests q1 rc=$?" | tee -a $O/tests_q1.log
tail -2 $O/tests_q1.log
for r in 1 2; do
  for l in libcellvit_amd_prev.so "" libcellvit_amd_q1.so; do
    echo "bench lib='$l'" >> $O/bench_ab.txt
    CVA_LIB=$l timeout 600 python bench.py --no-cpu-baseline --no-extras --allow-debug-env > $O/bench_last.log 2>&1; grep '^{' $O/bench_last.log >> $O/bench_ab.txt || tail -5 $O/bench_last.log
  done
done
python - <<'PY'
import json
for ln in open("gpurun_out/r04_z/bench_ab.txt"):
    if ln.startswith("{"):
        d = json.loads(ln); kc = d["kernel_classes"]
        print("   ", round(d["value"], 2), round(d["ms_per_step"], 1), {k.split("(")[0]: (round(v["total_ms_per_step"], 1), round(v["tflops"])) for k, v in kc.items()})
    else: print(ln.strip())
PY
