"""Summary of the Sinkhorn leg's rocprofv3 --kernel-trace run for the roofline: the fused pass's launches, how many of them
did the sweep (duration >= 20 % of the median: a launch that returns on the device-side stop word takes ~5 us), and the
mean duration of those.  usage: sink_profile_meta.py <trace_dir> <out.json>"""
import csv
import glob
import json
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from prof_summary import tree_stamp      # noqa: E402


def main():
    d, out = sys.argv[1], sys.argv[2]
    f = max(glob.glob(d + "/**/*kernel_trace.csv", recursive=True), key=os.path.getmtime)
    dur = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in csv.DictReader(open(f)) if "k_fused_pass<" in r["Kernel_Name"]]
    med = statistics.median(dur)
    swept = [x for x in dur if x >= 0.2 * med]
    rep = {**tree_stamp(), "kernel": "k_fused_pass", "launches": len(dur), "launches_that_swept": len(swept),
           "early_exit_launches": len(dur) - len(swept), "fused_pass_avg_us": sum(swept) / len(swept) / 1e3,
           "fused_pass_median_us": statistics.median(swept) / 1e3, "fused_pass_min_us": min(swept) / 1e3,
           "note": "profile runs set SPADOT_OT_SPEC_BATCHES=1 (no speculative batches, hence no launch that returns on the stop "
                   "word); any early exit left is filtered here"}
    json.dump(rep, open(out, "w"), indent=1)
    print(rep)


if __name__ == "__main__":
    main()
