#!/bin/bash
# round 6, call g: the tile loop of the product (cell_detection.run_tiles) with the previous batch's post-processing on a second stream, released by the current
# forward's full-resolution stage event — parity (tests/test_cli.py GPU legs: stand-in model + real engine, both orders the same cells), then the slide route
# (tools/bench_slide.py, 1024 tiles, 5.8e5 cells) with and without it, alternating, batch 16 and batch 64.
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06_g; mkdir -p $O
python -m pytest tests/test_cli.py tests/test_bench_ranks.py tests/test_gpu_stitch.py -m gpu -q > $O/pytest.log 2>&1; tail -3 $O/pytest.log
for b in 16 64; do
for v in serial overlap serial overlap; do
  if [ $v == serial ]; then F="--serial-postproc"; else F=""; fi
  timeout 300 python tools/bench_slide.py --tiles 1024 --batch $b $F > $O/s.json 2> $O/s.err
  python - "$v" "$b" $O/s.json <<'PY' | tee -a $O/slide_ab.txt
import json, sys
r = json.loads([l for l in open(sys.argv[3]) if l.startswith("{")][-1])
print("batch %s %-7s tile loop %.2f tiles/s (%.2f s)  slide total %.2f s  tail %.2f s  cells %d" % (sys.argv[2], sys.argv[1], r["tile_loop_tiles_per_s_rank0"], r["tile_loop_s"],
      r["slide_total_s"], r["tail_s"], r["cells_written"]))
PY
done
done
cp $O/s.json $O/slide_1024_b64_overlap.json
