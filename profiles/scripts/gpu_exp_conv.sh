#!/bin/bash
# A/B of the direct convolution's wave-group ping-pong (CVA_CONV_PP) on the decoder's layer shapes, then the conv tests.
OUT=gpurun_out/${1:-exp_conv}
mkdir -p $OUT
for pp in 0 1 0 1; do
  CVA_BUILD_FLAGS="-DCVA_CONV_PP=$pp" python -m cellvit_amd.build > /dev/null 2>&1
  echo "== CVA_CONV_PP=$pp" | tee -a $OUT/conv_ab.txt
  CONV_SHAPES=0,1,2,3,4,5,6,7,8 python tools/bench_conv.py 10 2>&1 | grep -v amdgpu.ids | tee -a $OUT/conv_ab.txt
done
python -m cellvit_amd.build > /dev/null 2>&1
python -m pytest tests/test_gpu_ops.py tests/test_gpu_forward.py -m gpu -q -x 2>&1 | tail -3 | tee -a $OUT/conv_ab.txt
python bench.py --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('bench f16', round(d['value'],2), {k[:10]:(round(v['tflops']), round(v['total_ms_per_step'],1)) for k,v in d['kernel_classes'].items()})" | tee -a $OUT/conv_ab.txt
