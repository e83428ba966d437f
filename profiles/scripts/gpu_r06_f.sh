#!/bin/bash
# round 6, call f: WHEN the second stream's post-processing chain is released (bench.py --pp-stage): 0 = behind the step's forward (the chain then meets the next
# step's encoder: persistent whole-CU GEMM workgroups it can only time-slice with), 1 = when the forward reaches its decoder, 2 = when its first branch reaches the
# full-resolution stages (short two-per-CU workgroups) — cv_stream_wait_stage.  Same call, product library, alternating.
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06_f; mkdir -p $O
for v in 0 1 2 0 1 2; do
  timeout 600 python bench.py --no-cpu-baseline --no-extras --pp-stage $v --steps 10 > $O/b.log 2>$O/b.err
  python - "$v" $O/b.log <<'PY' | tee -a $O/pp_stage_ab.txt
import json, sys
r = json.loads([l for l in open(sys.argv[2]) if l.startswith("{")][-1])
print("pp-stage=%s %.2f tiles/s %.1f ms/step; sequential: fwd %.1f pp %.1f; exposed %.1f ms" % (sys.argv[1], r["value"], r["ms_per_step"], r["stage_ms_sequential"]["forward"],
      r["stage_ms_sequential"]["postproc"], r["ms_per_step"] - r["stage_ms_sequential"]["forward"]),
      {k.split("(")[0]: round(v["total_ms_per_step"], 1) for k, v in r["kernel_classes"].items()})
PY
done
timeout 600 python bench.py --no-cpu-baseline --no-extras --no-overlap --steps 10 > $O/b.log 2>$O/b.err
python - "no-overlap" $O/b.log <<'PY' | tee -a $O/pp_stage_ab.txt
import json, sys
r = json.loads([l for l in open(sys.argv[2]) if l.startswith("{")][-1])
print("%s %.2f tiles/s %.1f ms/step; sequential: fwd %.1f pp %.1f" % (sys.argv[1], r["value"], r["ms_per_step"], r["stage_ms_sequential"]["forward"], r["stage_ms_sequential"]["postproc"]))
PY
for v in 0 2; do
  timeout 600 python bench.py --model vit256 --no-cpu-baseline --no-extras --pp-stage $v --steps 10 > $O/b.log 2>$O/b.err
  python - "vit256 pp-stage=$v" $O/b.log <<'PY' | tee -a $O/pp_stage_ab.txt
import json, sys
r = json.loads([l for l in open(sys.argv[2]) if l.startswith("{")][-1])
print("%s %.2f tiles/s %.1f ms/step; sequential: fwd %.1f pp %.1f" % (sys.argv[1], r["value"], r["ms_per_step"], r["stage_ms_sequential"]["forward"], r["stage_ms_sequential"]["postproc"]))
PY
done
