#!/bin/bash
# round 6, call d: LayerNorm folded into the block GEMMs (gamma / beta in the qkv / fc1 weights, row statistics from the proj / fc2 epilogues) — parity, then same-call A/B
# (production library, bench.py --no-ln-fold = the stand-alone LayerNorm kernels)
O=gpurun_out/r06_d; mkdir -p $O
python -m pytest tests/test_gpu_forward.py -q -x -k "fold or fp16" -s > $O/pytest_fold.log 2>&1; tail -12 $O/pytest_fold.log | cut -c1-300
python -m pytest tests/test_gpu_gemm8.py tests/test_gpu_ops.py -q -x > $O/pytest_ops.log 2>&1; tail -3 $O/pytest_ops.log
for v in "--no-ln-fold" "" "--no-ln-fold" ""; do
  timeout 600 python bench.py --no-cpu-baseline --no-extras --steps 6 $v > $O/b.log 2>$O/b.err
  python - "fold=$([ -z "$v" ] && echo 1 || echo 0)" $O/b.log <<'PY' | tee -a $O/bench_ab.txt
import json, sys
r = json.loads([l for l in open(sys.argv[2]) if l.startswith("{")][-1])
print(sys.argv[1], "%.2f tiles/s %.1f ms fwd %.1f flags %d" % (r["value"], r["ms_per_step"], r["stage_ms_sequential"]["forward"], r["config"]["engine_flags"]),
      {k.split("(")[0]: (v["launches"], round(v["total_ms_per_step"], 1), round(v["tflops"])) for k, v in r["kernel_classes"].items()}, r["parity"]["forward"]["hv_map"]["max_abs"], r["parity"]["forward"]["pass"])
PY
done
cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/$O/prof -o prof -- python $OLDPWD/bench.py --no-cpu-baseline --no-extras --no-postproc > /dev/null 2> $OLDPWD/$O/rocprof.err; cd $OLDPWD
find $O/prof -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats_fold.csv \; ; rm -rf $O/prof; head -14 $O/kernel_stats_fold.csv | cut -c1-160
