"""Per-kernel means of the FETCH_SIZE / WRITE_SIZE counters from two separate rocprofv3 --pmc passes
(the guide's HBM-traffic recipe: one counter per pass, no tracing domains besides the kernel trace).
usage: pmc_summary.py <fetch_dir> <write_dir> <out.csv>"""
import csv, glob, sys, collections

def means(d, counter):
    f = max(glob.glob(d + "/**/*counter_collection.csv", recursive=True), key=__import__("os").path.getmtime)
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == counter:
            acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return {k: (sum(v) / len(v), len(v)) for k, v in acc.items()}

fe, wr = means(sys.argv[1], "FETCH_SIZE"), means(sys.argv[2], "WRITE_SIZE")
with open(sys.argv[3], "w", newline="") as fh:
    w = csv.writer(fh)
    w.writerow(["kernel", "FETCH_SIZE_KB_mean_raw (x2 for 16B/lane streams on gfx950)", "n", "WRITE_SIZE_KB_mean", "n"])
    for k in sorted(set(fe) | set(wr)):
        a, b = fe.get(k, (0.0, 0)), wr.get(k, (0.0, 0))
        w.writerow([k, a[0], a[1], b[0], b[1]])
print("wrote", sys.argv[3])
