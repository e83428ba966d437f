#!/bin/bash
# A/B of the 8-phase kernel's DMA issue schedule (CVA_G8_SPLIT) on the SAM-H shapes at 32 tiles, fp16 and MX-fp8.
OUT=gpurun_out/${1:-exp_gemm}
mkdir -p $OUT
for split in 0 1 0 1; do
  CVA_BUILD_FLAGS="-DCVA_G8_SPLIT=$split" python -m cellvit_amd.build > /dev/null 2>&1
  echo "== CVA_G8_SPLIT=$split" | tee -a $OUT/gemm_ab.txt
  for shape in "131072 5120 1280" "131072 1280 5120" "131072 1280 1280" "131072 3840 1280"; do
    python tools/bench_gemm.py $shape 20 2>&1 | grep -v amdgpu.ids | tee -a $OUT/gemm_ab.txt
  done
  ACT=1 python tools/bench_gemm.py 131072 5120 1280 20 2>&1 | grep -v amdgpu.ids | sed 's/^/gelu: /' | tee -a $OUT/gemm_ab.txt
  RES=1 python tools/bench_gemm.py 131072 1280 5120 20 2>&1 | grep -v amdgpu.ids | sed 's/^/res: /' | tee -a $OUT/gemm_ab.txt
  python tools/bench_gemm_mx8.py 131072 5120 1280 20 2 1 2>&1 | grep -v amdgpu.ids | tee -a $OUT/gemm_ab.txt
  python tools/bench_gemm_mx8.py 131072 1280 5120 20 1 0 2>&1 | grep -v amdgpu.ids | tee -a $OUT/gemm_ab.txt
  python tools/bench_gemm_mx8.py 131072 3840 1280 20 0 0 2>&1 | grep -v amdgpu.ids | tee -a $OUT/gemm_ab.txt
done
python -m cellvit_amd.build > /dev/null 2>&1
python -m pytest tests/test_gpu_gemm8.py tests/test_gpu_fp8.py tests/test_gpu_ops.py -m gpu -q -x 2>&1 | tail -3 | tee -a $OUT/gemm_ab.txt
python bench.py --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('bench f16', round(d['value'],2), {k[:10]:round(v['tflops']) for k,v in d['kernel_classes'].items()})" | tee -a $OUT/gemm_ab.txt
python bench.py --no-cpu-baseline --dtype f8 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('bench f8', round(d['value'],2), {k[:10]:round(v['tflops']) for k,v in d['kernel_classes'].items()})" | tee -a $OUT/gemm_ab.txt
