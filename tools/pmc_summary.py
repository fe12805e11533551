"""Per-kernel means of hardware counters from separate rocprofv3 --pmc passes (the guide's HBM-traffic recipe: its
own run per counter group, no tracing domains besides the kernel trace).
usage: pmc_summary.py <out.csv> <pass_dir>:<COUNTER>[,<COUNTER>...] [<pass_dir>:<COUNTER>...] [--only substr]
Columns: one per counter (mean per launch, raw units as rocprofv3 reports them: FETCH_SIZE / WRITE_SIZE in KB --
FETCH_SIZE must be DOUBLED for 16-byte-per-lane streams on gfx950, MI355X_MICROARCH.md "HBM") + launches seen."""
import collections
import csv
import glob
import os
import sys


def means(d, counters):
    f = max(glob.glob(d + "/**/*counter_collection.csv", recursive=True), key=os.path.getmtime)
    acc = {c: collections.defaultdict(list) for c in counters}
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] in acc:
            acc[r["Counter_Name"]][r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return {c: {k: (sum(v) / len(v), len(v)) for k, v in acc[c].items()} for c in counters}


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    only = None
    if "--only" in sys.argv:
        only = sys.argv[sys.argv.index("--only") + 1]
        args = [a for a in args if a != only]
    out, passes = args[0], args[1:]
    cols, data = [], {}
    for p in passes:
        d, cs = p.rsplit(":", 1)
        cs = cs.split(",")
        m = means(d, cs)
        for c in cs:
            cols.append(c)
            data[c] = m[c]
    kernels = sorted(set(k for c in cols for k in data[c]))
    if only:
        kernels = [k for k in kernels if only in k]
    with open(out, "w", newline="") as fh:
        w = csv.writer(fh)
        hdr = ["kernel"]
        for c in cols:
            hdr += [c + "_mean" + ("_KB_raw(x2 for 16B/lane streams on gfx950)" if c == "FETCH_SIZE" else "_KB" if c == "WRITE_SIZE" else ""), "n"]
        if "TCC_HIT_sum" in cols and "TCC_MISS_sum" in cols:
            hdr.append("L2_hit_rate")
        w.writerow(hdr)
        for k in kernels:
            row = [k]
            for c in cols:
                a = data[c].get(k, (0.0, 0))
                row += [a[0], a[1]]
            if "TCC_HIT_sum" in cols and "TCC_MISS_sum" in cols:
                h, m_ = data["TCC_HIT_sum"].get(k, (0.0, 0))[0], data["TCC_MISS_sum"].get(k, (0.0, 0))[0]
                row.append(h / (h + m_) if h + m_ > 0 else "")
            w.writerow(row)
    print("wrote", out)


if __name__ == "__main__":
    main()
