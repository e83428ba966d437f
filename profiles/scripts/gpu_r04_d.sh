#!/bin/bash
# round 4, call D: after the attn_takes_vrm fix — whole GPU suite; rocprofv3 kernel stats of the bench step with V^T (CVA_NO_VRM=1) and with
# row-major V (which attention kernel pays the +1.7 ms of the class?)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04_d; mkdir -p $O
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -q -m gpu > $O/pytest.txt 2>&1; tail -5 $O/pytest.txt
ROOT=$(pwd)
export CVA_LIB=abl
for nv in 1 0; do
  (cd /tmp && CVA_NO_VRM=$nv rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/$O/prof_$nv -o prof -- python $ROOT/bench.py --allow-debug-env --no-cpu-baseline --no-extras --no-postproc --steps 3 > $ROOT/$O/bench_prof_$nv.json 2> $ROOT/$O/prof_$nv.err)
  find $O/prof_$nv -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats_novrm$nv.csv \;
  rm -rf $O/prof_$nv
done
python - <<'PY'
import csv
for nv in (1,0):
    rows=list(csv.DictReader(open(f'gpurun_out/r04_d/kernel_stats_novrm{nv}.csv')))
    print('NO_VRM', nv)
    for r in rows:
        if any(k in r['Name'] for k in ('attn','gemm8_kernel<1','pad_kv')):
            print(f"  {r['Name'][:70]:70s} calls {r['Calls']:>5s} avg_us {float(r['AverageNs'])/1e3:9.1f} total_ms {float(r['TotalDurationNs'])/1e6:8.1f}")
PY
