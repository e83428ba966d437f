#!/bin/bash
# round 5, call e: conv3x3_res_kernel against conv3x3_halo4_kernel on its layers (ablation library, CVA_CONV_RES = 0 / 1, same process order alternating),
# a rocprofv3 kernel-stats pass of the bench step, the streaming slide tail once more.
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05_e; mkdir -p $O
export TMPDIR=/tmp
{
for r in 1 2; do
  echo "halo4:"; CVA_LIB=abl CVA_CONV_RES=0 CONV_SHAPES=7,8 timeout 120 python tools/bench_conv.py 10 2>&1 | grep -v amdgpu
  echo "res  :"; CVA_LIB=abl CVA_CONV_RES=1 CONV_SHAPES=7,8 timeout 120 python tools/bench_conv.py 10 2>&1 | grep -v amdgpu
done
} | tee $O/conv_res_ab.txt
ROOT=$(pwd)
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/$O/prof -o prof -- python $ROOT/bench.py --no-cpu-baseline --no-extras --steps 3 > $ROOT/$O/bench_under_rocprof.json 2> $ROOT/$O/rocprof.err)
find $O/prof -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \;
find $O/prof -name "*kernel_trace.csv" -delete
rm -rf $O/prof
head -20 $O/kernel_stats.csv | cut -c1-200
timeout 400 python tools/bench_slide.py --tiles 1024 --batch 16 > $O/slide_stream.json 2> $O/slide_stream.err
cat $O/slide_stream.json
