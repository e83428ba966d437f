#!/bin/bash
# round 6, call j: soak of the tile loop's second-stream post-processing — a 2048-tile slide (1.2e6 cells) at the reference's batch of 8 (256 batches) and at 16, each in both
# orders; the two JSON documents must be byte-identical (sha256) between the orders
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06_j; mkdir -p $O
for b in 8 16; do
for v in overlap serial; do
  if [ $v == serial ]; then F="--serial-postproc"; else F=""; fi
  timeout 600 python tools/bench_slide.py --tiles 2048 --batch $b $F > $O/s_${b}_$v.json 2> $O/s.err
  python - "$v" "$b" $O/s_${b}_$v.json <<'PY' | tee -a $O/soak.txt
import json, sys
r = json.loads([l for l in open(sys.argv[3]) if l.startswith("{")][-1])
print("batch %s %-7s tile loop %.2f tiles/s  slide total %.2f s  tail %.2f s  cells %d  sha256 %s" % (sys.argv[2], sys.argv[1], r["tile_loop_tiles_per_s_rank0"], r["slide_total_s"], r["tail_s"],
      r["cells_written"], r["output_sha256_16"]))
PY
  rm -rf /tmp/cva_slide_*
done
done
