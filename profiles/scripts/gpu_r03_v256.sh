#!/bin/bash
# per-kernel time of the CellViT-256 step (configs[1])
OUT=gpurun_out/r03v256; mkdir -p $OUT
export TMPDIR=/tmp
ROOT=$(pwd)
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/$OUT/prof -o prof -- python $ROOT/bench.py --model vit256 --no-cpu-baseline --no-extras > $ROOT/$OUT/bench.json 2> $ROOT/$OUT/err.txt)
find $OUT/prof -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats_vit256.csv \;
rm -rf $OUT/prof
python - <<'PY'
import csv
rows=list(csv.DictReader(open('gpurun_out/r03v256/kernel_stats_vit256.csv')))
tot=sum(float(r['TotalDurationNs']) for r in rows)
for r in sorted(rows,key=lambda r:-float(r['TotalDurationNs']))[:16]:
    print(f"{r['Name'][:100]:100s} calls={r['Calls']:>5s} avg_us={float(r['AverageNs'])/1e3:9.1f} pct={100*float(r['TotalDurationNs'])/tot:5.1f}")
PY
