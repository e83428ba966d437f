#!/bin/bash
# round 4, call J2 (same passes, final tree with the balanced DMA schedule): PMC counters of this repository's linear GEMM and of the vendor kernel (hipBLASLt via torch.matmul) on the fc1 / fc2 shapes of a
# 64-tile step — VERDICT r03 item 1: "so the remaining gap is a counter difference, not a guess".  Separate --pmc passes, kernel-trace only.
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04_j2; mkdir -p $O
export TMPDIR=/tmp
ROOT=$(pwd)
python tools/bench_vendor_gemm.py 262144 6 2>&1 | grep -v amdgpu > $O/timing.txt; cat $O/timing.txt
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU" "SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU" "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_ANY SQ_INSTS_VALU_MFMA_MOPS_F16" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $ROOT/$O/pmc$i -o pmc -- python $ROOT/tools/bench_vendor_gemm.py 262144 2 > /dev/null 2> $ROOT/$O/pmc$i.err); echo "pass $i ($set) rc=$?"
done
python tools/pmc_insts.py $O/pmc1 $O/pmc2 $O/pmc3 $O/pmc4 $O/pmc5 $O/pmc6 > $O/kernel_counters.txt 2> $O/insts.err
rm -rf $O/pmc1 $O/pmc2 $O/pmc3 $O/pmc4 $O/pmc5 $O/pmc6
cut -c1-400 $O/kernel_counters.txt | head -12
