cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout -k 5 300 python -m pytest tests/test_gpu_gemm8.py tests/test_gpu_ops.py -m gpu -x -q -k "attention" < /dev/null 2>&1 | tail -3
for P in 1; do
  echo "== CVA_ATTNW_P=$P"
  CVA_ATTNW_P=$P timeout -k 5 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/aw_$P -o aw -- python tools/bench_attn.py 16 14 1 10 > gpurun_out/aw_$P.log 2>&1 < /dev/null
  f=$(find gpurun_out/aw_$P -name "*kernel_stats.csv" | head -1)
  if [ -n "$f" ]; then head -4 "$f" | cut -c1-170; fi
done
