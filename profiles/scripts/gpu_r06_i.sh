#!/bin/bash
# round 6, call i: HIP priority of the post-processing stream under the staged schedule (--pp-stage 2): normal against high; same call, alternating
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06_i; mkdir -p $O
for v in 0 -1 0 -1; do   # (--pp-priority: torch.cuda.Stream(dev, priority=v) for the post-processing stream; removed from bench.py after this call)
  timeout 600 python bench.py --no-cpu-baseline --no-extras --pp-priority $v --steps 10 > $O/b.log 2>$O/b.err
  python - "$v" $O/b.log <<'PY' | tee -a $O/pp_priority_ab.txt
import json, sys
r = json.loads([l for l in open(sys.argv[2]) if l.startswith("{")][-1])
print("pp-priority=%s %.2f tiles/s %.1f ms/step; sequential: fwd %.1f pp %.1f; exposed %.1f ms" % (sys.argv[1], r["value"], r["ms_per_step"], r["stage_ms_sequential"]["forward"],
      r["stage_ms_sequential"]["postproc"], r["ms_per_step"] - r["stage_ms_sequential"]["forward"]),
      {k.split("(")[0]: round(v["total_ms_per_step"], 1) for k, v in r["kernel_classes"].items()})
PY
done
for b in 32 16; do
  timeout 300 python tools/bench_slide.py --tiles 1024 --batch $b > $O/s.json 2> $O/s.err
  python - "$b" $O/s.json <<'PY' | tee -a $O/pp_priority_ab.txt
import json, sys
r = json.loads([l for l in open(sys.argv[2]) if l.startswith("{")][-1])
print("slide batch %s tile loop %.2f tiles/s  total %.2f s  tail %.2f s" % (sys.argv[1], r["tile_loop_tiles_per_s_rank0"], r["slide_total_s"], r["tail_s"]))
PY
done
