#!/bin/bash
# round 4, call A: (1) ds_read_b64_tr_b16 semantics + pitch timing; (2) are the 8-phase GEMM's epilogues a chip-wide store burst?
# start-delay stagger per XCD / per workgroup (CVA_GEMM_STAGGER, units of 10 ns) and sc1 / nt output stores (CVA_GEMM_DBG 8192 / 16384)
# at the production shapes of a 64-tile step (M = 262144); (3) a baseline bench line of this box.
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04_a; mkdir -p $O
tools/probes/_bin/probe_tr > $O/probe_tr.txt 2>&1; tail -12 $O/probe_tr.txt
export CVA_LIB=abl
run() { timeout 300 python tools/bench_gemm.py $1 $2 $3 10 2>&1 | grep -v amdgpu.ids | tail -2; }
{
for shape in "262144 5120 1280 1 0 4000" "262144 1280 5120 0 1 16000" "262144 1280 1280 0 0 4800"; do
  set -- $shape
  export ACT=$4 RES=$5 RACE=1
  echo "== $shape"
  CVA_GEMM_STAGGER=0 CVA_GEMM_DBG=0 run $1 $2 $3
  for s in $6 $(( $6 / 2 )) -$6 -$(( $6 / 2 )); do echo "stagger $s (x10 ns)"; CVA_GEMM_STAGGER=$s CVA_GEMM_DBG=0 run $1 $2 $3; done
  for d in 8192 16384; do echo "dbg $d"; CVA_GEMM_STAGGER=0 CVA_GEMM_DBG=$d run $1 $2 $3; done
  echo "stagger $6 + sc1"; CVA_GEMM_STAGGER=$6 CVA_GEMM_DBG=8192 run $1 $2 $3
  CVA_GEMM_STAGGER=0 CVA_GEMM_DBG=0 run $1 $2 $3
done
} > $O/bench_gemm_stagger.txt 2>&1
cat $O/bench_gemm_stagger.txt
unset CVA_LIB ACT RES RACE
python bench.py --no-cpu-baseline --no-extras > $O/bench_f16.json 2> $O/bench_f16.err; tail -c 1500 $O/bench_f16.json
