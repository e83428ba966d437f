"""Build the in-tree gfx950 shared library (hipcc cross-compiles without a GPU).

    python -m cellvit_amd.build            # -> cellvit_amd/libcellvit_amd.so
"""
from __future__ import annotations

import concurrent.futures as cf
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "csrc", "_obj")
LIB = os.path.join(HERE, "libcellvit_amd.so")
# The experiment flavour (-DCVA_ABLATION: CVA_* switches honoured, work-skipping instantiations present) has its own object
# directory and its own library name: the product library can never be linked from, or mistaken for, ablation objects.
# Experiment libraries (the ablation flavour, "previous commit" copies for same-call A/B runs) live OUTSIDE the package, under build/experimental/
# (git-ignored, travels to the GPU box): the package directory holds the product library and nothing else.
EXP_DIR = os.path.join(os.path.dirname(HERE), "build", "experimental")
OBJ_ABL = os.path.join(EXP_DIR, "_obj_abl")
LIB_ABL = os.path.join(EXP_DIR, "libcellvit_amd_abl.so")
ARCH = "gfx950"
FLAGS = ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Wno-unused-result"]


# Per-file flags.  -fno-slp-vectorize: hipcc's SLP vectoriser packs adjacent fp32 multiplies / adds into v_pk_mul_f32 / v_pk_add_f32, and a
# packed fp32 VALU instruction next to MFMAs (same wave or the SIMD's other wave) stalls the matrix pipe — measured
# (tools/probes/probe_mfma_valu_overlap.hip, profiles/r03_probe_mfma_valu_overlap.txt): plain v_fma_f32 fillers hide under the MFMAs, the
# packed form costs more than running the two streams one after the other.
FILE_FLAGS = {}


def _hipcc() -> str:
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: the cellvit_amd HIP extension cannot be built")


def sources(ablation: bool = False):
    """csrc/*.hip = the product; csrc/experiments/*.hip (kernels that lost their A/B and are kept for the record) join the experiment flavour only."""
    out = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))
    if ablation:
        exp = os.path.join(CSRC, "experiments")
        out += sorted(os.path.join(exp, f) for f in os.listdir(exp) if f.endswith(".hip"))
    return out


def _stale(out: str, deps) -> bool:
    if not os.path.exists(out):
        return True
    t = os.path.getmtime(out)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = True, ablation: bool = False) -> str:
    """ablation=True builds libcellvit_amd_abl.so with -DCVA_ABLATION (CVA_* experiment switches honoured; loaded only by
    tools/ that ask for it, never by bench.py / tests / the product).  Extra compiler flags (CVA_BUILD_FLAGS) are accepted for
    the ablation flavour only; the flags of either flavour are recorded in the library (cv_build_flags)."""
    hipcc = _hipcc()
    OBJ, LIB = (OBJ_ABL, LIB_ABL) if ablation else (globals()["OBJ"], globals()["LIB"])
    os.makedirs(OBJ, exist_ok=True)
    extra = os.environ.get("CVA_BUILD_FLAGS", "").split() if ablation else []
    flags = FLAGS + (["-DCVA_ABLATION"] if ablation else []) + extra
    flags = flags + ['-DCVA_BUILD_FLAGS_STR="' + " ".join(flags[1:]).replace('"', "'") + '"']
    stamp = os.path.join(OBJ, ".flags")          # (untracked: _obj*/ is git-ignored)
    stamp_text = " ".join(flags) + " | " + repr(sorted(FILE_FLAGS.items()))
    if not os.path.exists(stamp) or open(stamp).read() != stamp_text:
        force = True
        with open(stamp, "w") as f:
            f.write(stamp_text)
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    srcs = sources(ablation)
    headers.append(os.path.join(os.path.dirname(HERE), "include", "cellvit_amd.h"))
    jobs = []
    for src in srcs:
        obj = os.path.join(OBJ, os.path.basename(src)[:-4] + ".o")
        if force or _stale(obj, [src] + headers):
            jobs.append((src, obj))

    def cc(job):
        src, obj = job
        cmd = [hipcc] + flags + FILE_FLAGS.get(os.path.basename(src), []) + ["-c", src, "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed: " + " ".join(cmd) + "\n" + r.stdout + r.stderr)
        return src

    if jobs:
        with cf.ThreadPoolExecutor(max_workers=min(8, len(jobs))) as ex:
            for s in ex.map(cc, jobs):
                if verbose:
                    print("[cellvit_amd.build] compiled", os.path.basename(s), file=sys.stderr)
    objs = [os.path.join(OBJ, os.path.basename(s)[:-4] + ".o") for s in srcs]
    if force or jobs or _stale(LIB, objs):
        cmd = [hipcc, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", LIB] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed: " + " ".join(cmd) + "\n" + r.stdout + r.stderr)
        if verbose:
            print("[cellvit_amd.build] linked", LIB, file=sys.stderr)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv, ablation="--ablation" in sys.argv)
