#!/bin/bash
# round 3, call D: gemm2 (two workgroups per CU) correctness + A/B; slide bench with native writers; bench.py with extras
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r03_d; mkdir -p $O
export CVA_LIB=abl
for shape in "131072 5120 1280 1 0" "131072 1280 5120 0 1" "131072 1280 1280 0 0" "16384 3840 1280 0 0"; do
  set -- $shape
  for v in 0 1; do
    ACT=$4 RES=$5 RACE=2 CVA_GEMM2=$v timeout 300 python tools/bench_gemm.py $1 $2 $3 10 2>&1 | grep -v amdgpu.ids | tail -2 | sed "s/^/GEMM2=$v /"
  done
done > $O/bench_gemm2.txt 2>&1
cat $O/bench_gemm2.txt
unset CVA_LIB
timeout 600 python -m pytest tests/test_gpu_ops.py tests/test_cli.py -x -q -m gpu > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
timeout 900 python tools/bench_slide.py --tiles 1024 --batch 16 > $O/slide_1024.json 2> $O/slide_1024.err; cat $O/slide_1024.json
timeout 900 python bench.py --steps 5 --warmup 2 > $O/bench.json 2> $O/bench.err; tail -3 $O/bench.err; python - <<'PY'
import json
r=json.loads(open("gpurun_out/r03_d/bench.json").read().strip().splitlines()[-1])
print({k:r[k] for k in ("value","ms_per_step")}); print(r.get("extra")); print({k:(v["tflops"],v["total_ms_per_step"]) for k,v in r["kernel_classes"].items()})
PY
