#!/bin/bash
# round 4, call S: post-processing chain with four outputs per thread in the Sobel passes, word-wide 5x5 morphology and ballot-based run kernels (component / marker /
# instance statistics): bit-exact tests, then the chain alone (timing + per-kernel rocprof) and the end-to-end bench against the previous commit's library.
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04_s; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_postproc.py tests/test_evaluate.py tests/test_gpu_product_route.py tests/test_cli.py tests/test_gpu_forward.py -x -q -m gpu > $O/tests.log 2>&1; echo "tests rc=$?" | tee -a $O/tests.log
tail -5 $O/tests.log
for r in 1 2; do
  echo -n "prev: "; CVA_LIB=libcellvit_amd_prev.so timeout 600 python tools/bench_pp.py 64 5 2>&1 | grep -v amdgpu.ids
  echo -n "new : "; timeout 600 python tools/bench_pp.py 64 5 2>&1 | grep -v amdgpu.ids
done | tee $O/pp_ab.txt
R=$PWD; (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof -o pp -- python $R/tools/bench_pp.py 64 5 > $R/$O/prof.log 2>&1)
f=$(find $O/prof -name '*kernel_stats.csv' | head -1)
[ -n "$f" ] && cp "$f" $O/pp_kernel_stats.csv && head -40 $O/pp_kernel_stats.csv | cut -c1-150
rm -rf $O/prof
for r in 1 2; do
  for l in libcellvit_amd_prev.so ""; do
    echo "bench lib='$l'" >> $O/bench_ab.txt
    CVA_LIB=$l timeout 600 python bench.py --no-cpu-baseline --no-extras --allow-debug-env > $O/bench_last.log 2>&1; grep '^{' $O/bench_last.log >> $O/bench_ab.txt || tail -5 $O/bench_last.log
  done
done
python - <<'PY'
import json
for ln in open("gpurun_out/r04_s/bench_ab.txt"):
    if ln.startswith("{"):
        d = json.loads(ln)
        print("   ", round(d["value"], 2), round(d["ms_per_step"], 1), d["stage_ms_sequential"])
    else: print(ln.strip())
PY
