#!/bin/bash
# round 5, call a: the new parity cases (generic CellViT(768), SAM-L, partial batch under a larger geometry, fp8 engine with fp16 proj)
# and the default bench line with the new SURVEY §8d fields (parity gates, B = 1 / 8, K = 300 / 800 / 1500, CPU post-proc x n processes)
OUT=gpurun_out/r05_a
mkdir -p $OUT
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
python -m pytest tests -m gpu -q -s -k "vitgen or saml or partial_batch or fp16_proj or abi" > $OUT/pytest_new.log 2>&1; echo "pytest(new) rc=$?" > $OUT/rc.txt; tail -5 $OUT/pytest_new.log
( time python bench.py ) > $OUT/bench_f16.json 2> $OUT/bench_f16.err; echo "bench f16 rc=$?" >> $OUT/rc.txt; grep real $OUT/bench_f16.err >> $OUT/rc.txt
tail -3 $OUT/bench_f16.err
cat $OUT/rc.txt
