#!/bin/bash
# round 5, call o: what an item of attnwp_kernel (window attention, one workgroup per CU walking (window, head) items) is made of — ablation variants
# (CVA_ATTNWP_DBG, results wrong by construction) of the attention op on the production window shape, the kernel's average duration read from rocprofv3's kernel stats.
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05_o; mkdir -p $O
ROOT=$(pwd)
export TMPDIR=/tmp
for d in 0 1 2 4 6 8 14 15 16 31 32 63 0; do
  (cd /tmp && CVA_LIB=abl CVA_ATTNWP_DBG=$d timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/$O/prof_$d -o prof -- python $ROOT/tools/bench_attn.py 64 64 64 16 1280 14 10 > $ROOT/$O/run_$d.log 2>&1)
  f=$(find $O/prof_$d -name "*kernel_stats.csv" | head -1)
  echo -n "dbg $d: " | tee -a $O/attnwp_ablation.txt
  if [ -n "$f" ]; then grep "attnwp_kernel" "$f" | head -1 | awk -F'","|",|,"' '{print $1, "calls", $2, "avg_ns", $4}' | cut -c1-200 | tee -a $O/attnwp_ablation.txt; else echo "no stats" | tee -a $O/attnwp_ablation.txt; fi
  grep "us per call" $O/run_$d.log | sed 's/(qkv.*FLOPs/; FLOPs/' | tee -a $O/attnwp_ablation.txt
  rm -rf $O/prof_$d
done
