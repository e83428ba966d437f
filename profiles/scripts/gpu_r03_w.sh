#!/bin/bash
# per-kernel instruction mix / busy cycles (PMC passes, kernel-trace only) of the round-3 tree
OUT=gpurun_out/r03w2; mkdir -p $OUT
export TMPDIR=/tmp
ROOT=$(pwd)
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU" "SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU" "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY"; do
  i=$((i+1))
  (cd /tmp && timeout 900 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $ROOT/$OUT/pmc$i -o pmc -- python $ROOT/bench.py --no-cpu-baseline --no-extras --no-postproc --steps 1 --warmup 1 > /dev/null 2> $ROOT/$OUT/pmc$i.err); echo "pass $i rc=$?"
done
python tools/pmc_insts.py $OUT/pmc1 $OUT/pmc2 $OUT/pmc3 > $OUT/kernel_insts.txt 2> $OUT/insts.err
rm -rf $OUT/pmc1 $OUT/pmc2 $OUT/pmc3
head -12 $OUT/kernel_insts.txt | cut -c1-330
