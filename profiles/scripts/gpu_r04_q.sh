#!/bin/bash
# round 4, call Q: lean rows of the fused qkv projection's epilogue (qkv_rows) + opaque row pitches, against the previous commit's library on one box.
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04_q; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_ops.py tests/test_gpu_forward.py tests/test_gpu_gemm8.py tests/test_gpu_fp8.py -x -q -m gpu > $O/tests.log 2>&1; echo "tests rc=$?" | tee -a $O/tests.log
tail -3 $O/tests.log
for r in 1 2; do
  for l in libcellvit_amd_prev.so ""; do
    echo "bench lib='$l'" >> $O/bench_ab.txt
    CVA_LIB=$l timeout 600 python bench.py --no-cpu-baseline --no-extras --allow-debug-env > $O/bench_last.log 2>&1; grep '^{' $O/bench_last.log >> $O/bench_ab.txt || tail -5 $O/bench_last.log
  done
done
python - <<'PY'
import json
for ln in open("gpurun_out/r04_q/bench_ab.txt"):
    if ln.startswith("{"):
        d = json.loads(ln); kc = d["kernel_classes"]
        print("   ", round(d["value"], 2), round(d["ms_per_step"], 1), {k.split("(")[0]: (round(v["total_ms_per_step"], 1), round(v["tflops"])) for k, v in kc.items()})
    else: print(ln.strip())
PY
