#!/bin/bash
# round 6, call e: attnwp_kernel (14x14-window attention) — (1) rows of a window that lie in the image's padding are skipped per item (wave-uniform: query
# blocks are window rows), (2) static priority for one half of the workgroup, (3) the V image published behind the softmax instead of in front of it.
# Parity first (product library), then same-call A/B on the ablation library (CVA_ATTNW_DBG bits: 64 = no row skip (the kernel as it was), 128 = prio for waves 4-7,
# 512 = prio for waves 0-3, 256 = late V barrier), kernel duration from rocprofv3's kernel stats; then the bench line + kernel stats of this tree.
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06_e; mkdir -p $O
ROOT=$(pwd)
export TMPDIR=/tmp
python -m pytest tests/test_gpu_ops.py tests/test_gpu_forward.py tests/test_gpu_product_route.py tests/test_gpu_rccl_world1.py tests/test_gpu_fp8.py -q -x > $O/pytest.log 2>&1; tail -3 $O/pytest.log
for d in 64 0 128 256 384 768 640 64 0; do
  (cd /tmp && CVA_LIB=abl CVA_ATTNW_DBG=$d timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/$O/prof_$d -o prof -- python $ROOT/tools/bench_attn.py 64 64 64 16 1280 14 10 > $ROOT/$O/run_$d.log 2>&1)
  f=$(find $O/prof_$d -name "*kernel_stats.csv" | head -1)
  echo -n "dbg $d: " | tee -a $O/attnwp_ab.txt
  if [ -n "$f" ]; then grep "attnwp_kernel" "$f" | head -1 | awk -F'","|",|,"' '{print $1, "calls", $2, "avg_ns", $4}' | cut -c1-200 | tee -a $O/attnwp_ab.txt; else echo "no stats" | tee -a $O/attnwp_ab.txt; fi
  grep "us per call" $O/run_$d.log | sed 's/(qkv.*FLOPs/; FLOPs/' | tee -a $O/attnwp_ab.txt
  rm -rf $O/prof_$d
done
( time python bench.py ) > $O/bench_f16.json 2> $O/bench_f16.err; tail -3 $O/bench_f16.err
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/$O/prof -o prof -- python $ROOT/bench.py --no-cpu-baseline --no-extras > $ROOT/$O/bench_f16_under_rocprof.json 2> $ROOT/$O/rocprof.err)
find $O/prof -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats_f16.csv \;
rm -rf $O/prof
head -c 600 $O/bench_f16.json; echo; head -12 $O/kernel_stats_f16.csv | cut -c1-160
