#!/bin/bash
# round 4, call R: per-kernel times of the post-processing chain ALONE (64 tiles of 1024^2, 800 synthetic nuclei each; nothing else on the GPU): rocprofv3 kernel trace of
# tools/bench_pp.py — where the 11.5 ms that are not the ordered flood go.
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04_r; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python tools/bench_pp.py 64 5 2>&1 | grep -v amdgpu.ids | tee $O/pp.txt
R=$PWD; (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof -o pp -- python $R/tools/bench_pp.py 64 5 > $R/$O/prof.log 2>&1)
f=$(find $O/prof -name '*kernel_stats.csv' | head -1)
[ -n "$f" ] && cp "$f" $O/pp_kernel_stats.csv && head -50 $O/pp_kernel_stats.csv | cut -c1-200
rm -rf $O/prof
