"""Fabric-side bytes per launch of the GEMM-class kernels from two rocprofv3 PMC passes -> profiles/traffic_latest.json.

    rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_f -o pmc -- \
        python bench.py --no-cpu-baseline --no-postproc --steps 1 --warmup 1
    rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_w -o pmc -- (same command)
    python tools/pmc_traffic.py gpurun_out/pmc_f gpurun_out/pmc_w

FETCH_SIZE / WRITE_SIZE are reported in KB; FETCH_SIZE is doubled as MI355X_MICROARCH.md prescribes for gfx950 (the
128-byte requests of 16-byte-per-lane streaming reads are tallied at 64 B).  Separate passes: never combined with other
trace domains."""
import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLASSES = [  # (key in the json, substring of the kernel name); gemm8_kernel<OMODE, TRANS, ABL, F8, CV3>
    ("mx8", "gemm8_kernel<0, 1, 0, 1, 0>"), ("mx8", "gemm8_kernel<1, 1, 0, 1, 0>"), ("mx8", "gemm8_kernel<3, 1, 0, 1, 0>"),
    ("conv", "gemm8_kernel<0, 1, 0, 0, 1>"),
    ("linear", "gemm8_kernel<0"), ("qkv", "gemm8_kernel<1"), ("convT", "gemm8_kernel<2"), ("conv", "conv3x3_halo_kernel"),
    ("conv_halo4", "conv3x3_halo4_kernel"), ("conv_res", "conv3x3_res_kernel"),
    ("attn_global", "attn2d_kernel"), ("attn_global", "attn2_kernel"), ("attn_win", "attnwp_kernel"), ("layernorm_mx8", "layernorm_mx8_kernel"),
    ("layernorm", "layernorm_kernel"), ("layernorm_add", "layernorm_add_kernel"),
]
BENCH_KEYS = {"linear": "gemm_linear(proj/fc1/fc2/patch/neck)", "qkv": "gemm_qkv", "conv": "conv3x3_implicit_gemm",
              "convT": "convT2x2_gemm", "mx8": "gemm_mx8(qkv/fc1/fc2, MX-fp8)"}


def per_launch(directory, counter):
    files = glob.glob(os.path.join(directory, "**", "*counter_collection.csv"), recursive=True)
    if not files:
        raise SystemExit(f"no counter_collection.csv under {directory}")
    tot = collections.defaultdict(float)
    cnt = collections.defaultdict(set)
    for n, r in enumerate(csv.DictReader(open(files[0]))):
        if r["Counter_Name"] != counter:
            continue
        name = r["Kernel_Name"]
        for key, sub in CLASSES:
            if sub in name and not (key == "layernorm" and "layernorm_add" in name):
                tot[key] += float(r["Counter_Value"])
                cnt[key].add(r.get("Dispatch_Id", n))
                break
    return {k: tot[k] / max(len(cnt[k]), 1) for k in tot}


def main():
    fdir, wdir = sys.argv[1], sys.argv[2]
    fetch = per_launch(fdir, "FETCH_SIZE")
    write = per_launch(wdir, "WRITE_SIZE")
    out = {"_how": __doc__.split("\n\n")[1].replace("\n", " ").strip() + "  " + __doc__.split("\n\n")[2].replace("\n", " ").strip()}
    detail = {}
    for key in dict.fromkeys(k for k, _ in CLASSES):
        if key in fetch or key in write:
            f2 = 2.0 * fetch.get(key, 0.0) * 1024.0
            w = write.get(key, 0.0) * 1024.0
            detail[key] = {"fetch_x2": round(f2 / 2 ** 20, 1), "write": round(w / 2 ** 20, 1)}
            if key in BENCH_KEYS:
                out[BENCH_KEYS[key]] = f2 + w
    out["_detail_MiB_per_launch"] = detail
    path = os.path.join(ROOT, "profiles", sys.argv[3] if len(sys.argv) > 3 else "traffic_latest.json")
    # provenance (bench.py prints it next to roofline.traffic): tiles per step of the profiled command, commit of the tree
    import subprocess
    out["tiles_per_step"] = int(os.environ.get("PMC_TILES_PER_STEP", "64"))       # bench.py default --batch unless the caller says otherwise
    try:
        out["commit"] = subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip() or None
    except OSError:
        out["commit"] = None
    json.dump(out, open(path, "w"), indent=1)
    print(json.dumps(detail, indent=1))


if __name__ == "__main__":
    main()
