#!/bin/bash
# round 4, call E: window layers row-major V / global layers V^T.  (1) op + forward tests; (2) window kernel: register-prefetch form
# (production, VRM = 1) against the LDS-DMA form (CVA_VRM_DMA=1) and against V^T (CVA_NO_VRM=1), rocprofv3 kernel stats, same call;
# (3) bench lines.
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04_e; mkdir -p $O
export TMPDIR=/tmp
timeout 1800 python -m pytest tests/test_gpu_ops.py tests/test_gpu_gemm8.py tests/test_gpu_forward.py tests/test_gpu_product_route.py tests/test_gpu_fp8.py -q -m gpu > $O/pytest.txt 2>&1; tail -4 $O/pytest.txt
ROOT=$(pwd)
export CVA_LIB=abl
for v in "CVA_NO_VRM=1" "CVA_VRM_DMA=0" "CVA_VRM_DMA=1"; do
  tag=$(echo $v | tr '=' '_')
  (cd /tmp && env $v rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/$O/prof_$tag -o prof -- python $ROOT/bench.py --allow-debug-env --no-cpu-baseline --no-extras --no-postproc --steps 3 > $ROOT/$O/bench_prof_$tag.json 2> $ROOT/$O/prof_$tag.err)
  find $O/prof_$tag -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats_$tag.csv \;
  rm -rf $O/prof_$tag
done
python - <<'PY'
import csv, glob
for f in sorted(glob.glob('gpurun_out/r04_e/kernel_stats_*.csv')):
    rows=list(csv.DictReader(open(f)))
    print(f)
    for r in rows:
        if any(k in r['Name'] for k in ('attn','gemm8_kernel<1')):
            print(f"  {r['Name'][:70]:70s} calls {r['Calls']:>5s} avg_us {float(r['AverageNs'])/1e3:9.1f} total_ms {float(r['TotalDurationNs'])/1e6:8.1f}")
PY
summ() { python - "$1" <<'PY'
import json, sys
r=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1], round(r['value'],2), 'tiles/s', round(r['ms_per_step'],1), 'ms', {k.split('(')[0]: (round(v['tflops']), round(v['total_ms_per_step'],1)) for k,v in r['kernel_classes'].items()})
PY
}
for rep in 1 2; do
  for v in "CVA_NO_VRM=1" "CVA_VRM_DMA=0" "CVA_VRM_DMA=1"; do
    tag=$(echo $v | tr '=' '_')
    env $v python bench.py --allow-debug-env --no-cpu-baseline --no-extras --steps 10 > $O/bench_${tag}_$rep.json 2> $O/bench_${tag}_$rep.err; summ $O/bench_${tag}_$rep.json
  done
done
