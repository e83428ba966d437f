"""Per-kernel instruction-mix summary from rocprofv3 --pmc passes (VALU / MFMA / LDS instruction counts, busy cycles).
    cd /tmp && rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA --output-format csv -d <dir> -o pmc -- python bench.py --steps 1 --warmup 1 ...
    python tools/pmc_insts.py <dir> [<dir2> ...]  > profiles/rNN_kernel_insts.txt
Counters are summed over all launches of a kernel (name truncated at the template arguments' end) and divided by the launch count."""
import collections
import csv
import glob
import os
import re
import sys


def short(name):
    name = re.sub(r"^void\s+", "", name)
    name = re.sub(r"cva::\(anonymous namespace\)::|cva::", "", name)
    m = re.match(r"([A-Za-z0-9_]+(<[^()]*>)?)", name)
    return m.group(1) if m else name[:60]


def main():
    tot = collections.defaultdict(lambda: collections.defaultdict(float))
    cnt = collections.defaultdict(lambda: collections.defaultdict(set))
    for d in sys.argv[1:]:
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            for n, r in enumerate(csv.DictReader(open(f))):
                k = short(r["Kernel_Name"])
                c = r["Counter_Name"]
                tot[k][c] += float(r["Counter_Value"])
                cnt[k][c].add(r.get("Dispatch_Id", n))
    counters = sorted({c for k in tot for c in tot[k]})
    print("kernel".ljust(58) + "launches " + " ".join(c.rjust(22) for c in counters) + "   VALU/MFMA")
    rows = []
    for k in tot:
        n = max(len(s) for s in cnt[k].values())
        vals = [tot[k].get(c, 0.0) / max(len(cnt[k].get(c, ())), 1) for c in counters]
        rows.append((sum(tot[k].get("SQ_INSTS_VALU", 0.0) for _ in (0,)), k, n, vals))
    for _, k, n, vals in sorted(rows, reverse=True)[:24]:
        d = dict(zip(counters, vals))
        ratio = d.get("SQ_INSTS_VALU", 0.0) / d["SQ_INSTS_MFMA"] if d.get("SQ_INSTS_MFMA") else float("nan")
        print(k[:57].ljust(58) + f"{n:8d} " + " ".join(f"{v:22.4g}" for v in vals) + f"   {ratio:9.2f}")


if __name__ == "__main__":
    main()
