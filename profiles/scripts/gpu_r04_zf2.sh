#!/bin/bash
# round 4, call ZF2: balanced DMA schedule in the fp8 K loop, engine level: one fp8 forward-statistics test, then the fp8 bench against the previous commit's library.
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04_zf2; mkdir -p $O
timeout 200 python -m pytest tests/test_gpu_fp8.py -x -q -m gpu -k "forward_fp8_error_statistics and samh_256" > $O/tests.log 2>&1; echo "tests rc=$?" | tee -a $O/tests.log
tail -2 $O/tests.log
for l in libcellvit_amd_prev.so "" libcellvit_amd_prev.so ""; do
  echo "bench f8 lib='$l'" >> $O/bench_ab.txt
  CVA_LIB=$l timeout 200 python bench.py --dtype f8 --no-cpu-baseline --no-extras --allow-debug-env --steps 3 --warmup 1 2>/dev/null | grep '^{' >> $O/bench_ab.txt
done
python - <<'PY'
import json
for ln in open("gpurun_out/r04_zf2/bench_ab.txt"):
    if ln.startswith("{"):
        d = json.loads(ln); kc = d["kernel_classes"]
        print("   ", round(d["value"], 2), round(d["ms_per_step"], 1), {k.split("(")[0]: (round(v["total_ms_per_step"], 1), round(v["tflops"])) for k, v in kc.items()})
    else: print(ln.strip())
PY
