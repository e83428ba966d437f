#!/bin/bash
# round 3, call C: device stitch tests, new forward goldens, full GPU suite, slide benchmark
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r03_c; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_stitch.py tests/test_cli.py tests/test_gpu_forward.py -x -q -m gpu -s > $O/pytest_new.txt 2>&1
tail -5 $O/pytest_new.txt; grep "^\[stitch\]\|^\[cli\]\|vit256_1024" $O/pytest_new.txt
timeout 1800 python -m pytest tests -x -q -m gpu > $O/pytest_all.txt 2>&1
tail -4 $O/pytest_all.txt
timeout 900 python tools/bench_slide.py --tiles 256 --batch 16 > $O/slide_256.json 2> $O/slide_256.err; tail -2 $O/slide_256.err; cat $O/slide_256.json
timeout 900 python tools/bench_slide.py --tiles 1024 --batch 16 > $O/slide_1024.json 2> $O/slide_1024.err; cat $O/slide_1024.json
timeout 900 python tools/bench_slide.py --tiles 256 --batch 16 --ranks 2 > $O/slide_256_r2.json 2> $O/slide_256_r2.err; tail -3 $O/slide_256_r2.err; cat $O/slide_256_r2.json
