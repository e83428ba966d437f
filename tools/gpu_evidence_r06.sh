#!/bin/bash
# Round-6 evidence call: one GPU box, one call, the final tree -> gpurun_out/$1 (the summaries that should be judged are copied into profiles/).
#   tools/gpu_evidence_r06.sh r06_z [tests]
OUT=gpurun_out/${1:-r06_z}
mkdir -p $OUT
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
if [ "$2" == "tests" ]; then
  python -m pytest tests -m gpu -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?" > $OUT/rc.txt; tail -4 $OUT/pytest.log
fi
( time python bench.py ) > $OUT/bench_f16.json 2> $OUT/bench_f16.err; echo "bench f16 rc=$?" >> $OUT/rc.txt; grep real $OUT/bench_f16.err >> $OUT/rc.txt
python bench.py --dtype f8 --no-extras > $OUT/bench_f8.json 2> $OUT/bench_f8.err; echo "bench f8 rc=$?" >> $OUT/rc.txt
python bench.py --model vit256 --no-cpu-baseline > $OUT/bench_vit256.json 2> $OUT/bench_vit256.err; echo "bench vit256 rc=$?" >> $OUT/rc.txt
ROOT=$(pwd)
export PMC_TILES_PER_STEP=64
for dt in f16 f8; do
  (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/$OUT/prof_$dt -o prof -- python $ROOT/bench.py --dtype $dt --no-cpu-baseline --no-extras > $ROOT/$OUT/bench_${dt}_under_rocprof.json 2> $ROOT/$OUT/rocprof_$dt.err)
  (cd /tmp && rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $ROOT/$OUT/pmc_f_$dt -o pmc -- python $ROOT/bench.py --dtype $dt --no-cpu-baseline --no-extras --no-postproc --steps 1 --warmup 1 > /dev/null 2> $ROOT/$OUT/pmc_f_$dt.err)
  (cd /tmp && rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $ROOT/$OUT/pmc_w_$dt -o pmc -- python $ROOT/bench.py --dtype $dt --no-cpu-baseline --no-extras --no-postproc --steps 1 --warmup 1 > /dev/null 2> $ROOT/$OUT/pmc_w_$dt.err)
  find $OUT/prof_$dt -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats_$dt.csv \;
  python tools/pmc_traffic.py $OUT/pmc_f_$dt $OUT/pmc_w_$dt ../$OUT/traffic_$dt.json > $OUT/traffic_$dt.txt 2>&1
  rm -rf $OUT/pmc_f_$dt $OUT/pmc_w_$dt
  find $OUT/prof_$dt -name "*kernel_trace.csv" -delete
  rm -rf $OUT/prof_$dt
done
# instruction mix / busy cycles of the fp16 step (three PMC passes, kernel-trace only)
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU" "SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY"; do
  i=$((i+1))
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $ROOT/$OUT/pmci$i -o pmc -- python $ROOT/bench.py --no-cpu-baseline --no-extras --no-postproc --steps 1 --warmup 1 > /dev/null 2> $ROOT/$OUT/pmci$i.err); echo "insts pass $i rc=$?" >> $OUT/rc.txt
done
python tools/pmc_insts.py $OUT/pmci1 $OUT/pmci2 $OUT/pmci3 > $OUT/kernel_insts.txt 2> $OUT/insts.err
rm -rf $OUT/pmci1 $OUT/pmci2 $OUT/pmci3
python tools/bench_slide.py --tiles 1024 --batch 16 > $OUT/slide_1024_b16.json 2> $OUT/slide_1024_b16.err
python tools/bench_slide.py --tiles 1024 --batch 64 > $OUT/slide_1024_b64.json 2> $OUT/slide_1024_b64.err
python tools/bench_slide.py --tiles 1024 --batch 16 --ranks 2 --backend gloo > $OUT/slide_1024_b16_2ranks.json 2> $OUT/slide_1024_b16_2ranks.err
# the N-rank path of bench.py itself, as far as one GPU allows: two ranks sharing cuda:0 over gloo, slide leg on the run's process group
python bench.py --gpus 2 --backend gloo --share-gpu --batch 16 --steps 3 --warmup 1 2> $OUT/bench_2ranks_gloo.err | grep '^{' > $OUT/bench_2ranks_gloo.json
# the reference CLI's default batch (8 tiles per step): classes of that step
python bench.py --batch 8 --no-extras --no-cpu-baseline --steps 20 --warmup 4 > $OUT/bench_f16_batch8.json 2> $OUT/bench_f16_batch8.err
# the reference CLI's default batch through the real tile loop, and the loop's one-stream form beside it
python tools/bench_slide.py --tiles 1024 --batch 8 > $OUT/slide_1024_b8.json 2> $OUT/slide_1024_b8.err
python tools/bench_slide.py --tiles 1024 --batch 8 --serial-postproc > $OUT/slide_1024_b8_serial.json 2> $OUT/slide_1024_b8_serial.err
python tools/bench_slide.py --tiles 1024 --batch 16 --serial-postproc > $OUT/slide_1024_b16_serial.json 2> $OUT/slide_1024_b16_serial.err
python tools/bench_slide.py --tiles 1024 --batch 16 --model vit256 > $OUT/slide_1024_b16_vit256.json 2> $OUT/slide_1024_b16_vit256.err
# rounds 2-5's release point of the post-processing stream, for the record
python bench.py --pp-stage 0 --no-extras --no-cpu-baseline > $OUT/bench_f16_pp_stage0.json 2> $OUT/bench_f16_pp_stage0.err
python -c "import __graft_entry__ as g; g.smoke(); print('smoke OK')" > $OUT/smoke.txt 2>&1; tail -1 $OUT/smoke.txt >> $OUT/rc.txt
cat $OUT/rc.txt; cat $OUT/slide_1024_b16.json $OUT/slide_1024_b64.json $OUT/slide_1024_b16_2ranks.json; head -8 $OUT/kernel_insts.txt | cut -c1-250
