#!/bin/bash
# round 4, call W: what the dependent global load of the neighbour's distance value costs the ordered flood per pop (ablation library, CVA_PP_DBG=1 replaces the load by
# arithmetic: wrong flood order, timing only) — the upper bound of what an LDS-resident distance tile for the large components could buy.
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04_w; mkdir -p $O
export TMPDIR=/tmp CVA_LIB=abl
for d in 0 1 0 1; do
  R=$PWD; (cd /tmp && CVA_PP_DBG=$d timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof$d -o pp -- python $R/tools/bench_pp.py 64 5 > $R/$O/prof.log 2>&1)
  f=$(find $O/prof$d -name '*kernel_stats.csv' | head -1)
  echo "CVA_PP_DBG=$d: $(grep B= $O/prof.log)  $(grep k_flood $f | awk -F, '{print "k_flood avg ns", $4, "min", $6, "max", $7}')"
  rm -rf $O/prof$d
done | tee $O/flood_dist_load.txt
