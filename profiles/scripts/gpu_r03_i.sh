#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r03_i; mkdir -p $O
export TMPDIR=/tmp
ROOT=$(pwd)
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/$O/prof -o prof -- python $ROOT/tools/bench_slide.py --tiles 512 --batch 64 > $ROOT/$O/slide.json 2> $ROOT/$O/slide.err)
cat $O/slide.json
find $O/prof -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \;
find $O/prof -name "*kernel_trace.csv" -delete
python - <<'PY'
import csv
rows=list(csv.DictReader(open("gpurun_out/r03_i/kernel_stats.csv")))
tot=sum(float(r['TotalDurationNs']) for r in rows)
print("total kernel ms", tot/1e6)
for r in sorted(rows,key=lambda r:-float(r['TotalDurationNs']))[:28]:
    print(f"{r['Name'][:60]:60s} calls {int(r['Calls']):6d} avg {float(r['AverageNs'])/1e3:9.1f} us total {float(r['TotalDurationNs'])/1e6:8.1f} ms")
PY
