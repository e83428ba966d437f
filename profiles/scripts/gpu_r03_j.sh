#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r03_j; mkdir -p $O
timeout 900 python -m pytest tests/test_cli.py -x -q -m gpu > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
timeout 900 python tools/bench_slide.py --tiles 1024 --batch 16 > $O/slide_1024_b16.json 2> $O/slide_1024_b16.err; cat $O/slide_1024_b16.json
timeout 900 python tools/bench_slide.py --tiles 1024 --batch 64 --slides 2 > $O/slide_1024_b64.json 2> $O/slide_1024_b64.err; cat $O/slide_1024_b64.json
