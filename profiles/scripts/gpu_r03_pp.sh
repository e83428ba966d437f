#!/bin/bash
OUT=gpurun_out/r03pp; mkdir -p $OUT
export TMPDIR=/tmp
ROOT=$(pwd)
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/$OUT/prof -o prof -- python $ROOT/tools/bench_pp.py > $ROOT/$OUT/pp.txt 2> $ROOT/$OUT/err.txt)
find $OUT/prof -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats_pp.csv \;
rm -rf $OUT/prof
tail -3 $OUT/pp.txt
python - <<'PY'
import csv
rows=list(csv.DictReader(open('gpurun_out/r03pp/kernel_stats_pp.csv')))
tot=sum(float(r['TotalDurationNs']) for r in rows)
for r in sorted(rows,key=lambda r:-float(r['TotalDurationNs']))[:24]:
    print(f"{r['Name'][:80]:80s} calls={r['Calls']:>5s} avg_us={float(r['AverageNs'])/1e3:9.1f} pct={100*float(r['TotalDurationNs'])/tot:5.1f}")
PY
