#!/bin/bash
# round 5, call d: (1) the persistent 3x3 convolution with the filter resident in LDS (conv3x3_res_kernel: Cin 32 / 64 -> 64 at full resolution, incl. the
# fused heads): op / forward / product-route tests, previous library against this one on the SAM-H and CellViT-256 steps; (2) window attention kernel of
# call c against the one before it on the attention op (same call); (3) the streaming slide tail against the batch route on a 1024-tile slide.
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05_d; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_forward.py tests/test_gpu_product_route.py tests/test_cli.py -x -q -m gpu -k "conv or samh or vit256 or product or route or cli" > $O/tests.log 2>&1; echo "tests rc=$?" | tee -a $O/tests.log
tail -3 $O/tests.log
for r in 1 2; do
  CVA_LIB=libcellvit_amd_prev.so timeout 300 python bench.py --allow-debug-env --no-extras --no-cpu-baseline --steps 6 > $O/bench_prev_$r.json 2> $O/bench_prev_$r.err
  timeout 300 python bench.py --no-extras --no-cpu-baseline --steps 6 > $O/bench_new_$r.json 2> $O/bench_new_$r.err
done
CVA_LIB=libcellvit_amd_prev.so timeout 300 python bench.py --allow-debug-env --model vit256 --no-extras --no-cpu-baseline --steps 6 > $O/vit256_prev.json 2> $O/vit256_prev.err
timeout 300 python bench.py --model vit256 --no-extras --no-cpu-baseline --steps 6 > $O/vit256_new.json 2> $O/vit256_new.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r05_d/*_*.json')):
    try:
        r=json.load(open(f)); k=r['kernel_classes']
        print(f.split('/')[-1], round(r['value'],2), 'fwd', round(r['stage_ms_sequential']['forward'],1), 'conv', round(k['conv3x3_implicit_gemm']['total_ms_per_step'],2), round(k['conv3x3_implicit_gemm']['tflops']), 'attn', round(k['attention']['total_ms_per_step'],2))
    except Exception as e: print(f, 'ERR', e)
PY
{
for r in 1 2 3; do
  echo -n "oldwin: "; CVA_LIB=libcellvit_amd_oldwin.so timeout 120 python tools/bench_attn.py 64 64 64 16 1280 14 20 2>&1 | grep -v amdgpu | tail -1
  echo -n "new   : "; timeout 120 python tools/bench_attn.py 64 64 64 16 1280 14 20 2>&1 | grep -v amdgpu | tail -1
done
} | tee $O/attn_win_ab.txt
timeout 400 python tools/bench_slide.py --tiles 1024 --batch 16 > $O/slide_stream.json 2> $O/slide_stream.err
timeout 400 python tools/bench_slide.py --tiles 1024 --batch 16 --batch-tail > $O/slide_batch.json 2> $O/slide_batch.err
timeout 400 python tools/bench_slide.py --tiles 1024 --batch 16 --ranks 2 > $O/slide_stream_2ranks.json 2> $O/slide_stream_2ranks.err
cat $O/slide_stream.json $O/slide_batch.json $O/slide_stream_2ranks.json; tail -3 $O/slide_stream.err
