#!/usr/bin/env python3
"""Copy the summaries of a tools/gpu_evidence_r06.sh call from gpurun_out/<tag>/ into profiles/<tag>_* (tracked) and refresh
profiles/traffic_latest.json / traffic_f8.json (what bench.py reads for roofline.traffic) with the commit they were collected at.

    python tools/collect_evidence.py r06_z [commit]"""
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
commit = sys.argv[2] if len(sys.argv) > 2 else subprocess.check_output(["git", "rev-parse", "--short", "HEAD"], cwd=ROOT, text=True).strip()
src, dst = os.path.join(ROOT, "gpurun_out", tag), os.path.join(ROOT, "profiles")
names = {"bench_f16.json": "bench_f16.json", "bench_f8.json": "bench_f8.json", "bench_vit256.json": "bench_vit256.json",
         "bench_f16_batch8.json": "bench_f16_batch8.json", "bench_f16_under_rocprof.json": "bench_f16_under_rocprof.json",
         "bench_f8_under_rocprof.json": "bench_f8_under_rocprof.json", "kernel_stats_f16.csv": "bench_f16_kernel_stats.csv",
         "kernel_stats_f8.csv": "bench_f8_kernel_stats.csv", "kernel_insts.txt": "kernel_insts.txt", "pytest.log": "pytest_gpu_tail.txt",
         "slide_1024_b16.json": "slide_1024_batch16.json", "slide_1024_b64.json": "slide_1024_batch64.json",
         "slide_1024_b16_2ranks.json": "slide_1024_batch16_2ranks_gloo.json", "bench_2ranks_gloo.json": "bench_2ranks_gloo_share_gpu.json",
         "probe_mfma_shape.txt": "probe_mfma_shape.txt"}
for a, b in names.items():
    p = os.path.join(src, a)
    if os.path.exists(p) and os.path.getsize(p):
        if a.endswith(".json"):      # keep the JSON line only (gloo prints a banner on stdout)
            lines = [ln for ln in open(p) if ln.startswith("{")]
            if not lines:
                continue
            open(os.path.join(dst, f"{tag}_{b}"), "w").write(lines[-1])
        else:
            shutil.copy(p, os.path.join(dst, f"{tag}_{b}"))
for dt, out in (("f16", "traffic_latest.json"), ("f8", "traffic_f8.json")):
    p = os.path.join(src, f"traffic_{dt}.json")
    if os.path.exists(p):
        t = json.load(open(p))
        t["commit"] = commit
        t["evidence_call"] = tag
        json.dump(t, open(os.path.join(dst, out), "w"), indent=1)
print("copied", tag, "at commit", commit)
