"""Per-step kernel breakdown from a rocprofv3 --kernel-trace csv of `bench.py --leg train` (the last
`steps` graph replays are taken as the timed region).
usage: prof_summary.py <dir> [steps] [top_n] [families.json]
With a fourth argument the per-family totals (us per step, launches per step) are also written as JSON: that file,
committed under profiles/, is what bench.py's `roofline_train` block replays."""
import collections
import csv
import glob
import json
import os
import sys


def family(name):
    """Kernel family of one launch, by name (Tensile names: BBS/BSS/_B_ = bf16 inputs, _DB_ = fp64, _S_/SB = fp32)."""
    n = name
    if "k_gat_" in n or "k_tail_" in n:          # csrc/gat_mfma.hip, model_kernels.hip and the aggregate-first last layer (gat_tail.hip)
        return "gat_edge"
    if "k_dgemm_small" in n:
        return "gemm_f64_library"       # (the same family as the library's fp64 products it stands in for when switched on)
    if "k_gemm_bf16" in n or "k_gemm_wgrad_bf16" in n or "k_wgrad_reduce" in n or "k_gemm_tn_bf16" in n or "k_gemm_tail_reduce" in n:
        return "gemm_bf16_own"          # csrc/gemm_bf16.hip, csrc/gemm_wgrad_bf16.hip (with its partial-sum launch)
    if n.startswith("Cijk_") or n.startswith("Custom_Cijk") or "rocblas_gem" in n or "gemv" in n.lower():
        if "_DB_" in n or "double" in n:
            return "gemm_f64_library"
        if "_BBS_" in n or "_BSS_" in n:          # bf16 operands (bf16 or fp32 result)
            return "gemm_bf16_library"
        return "gemm_f32_library"                 # "_S_B_Bias...": fp32 operands
    if "k_spd_sweep" in n:
        return "svgp_sweep"
    if "k_adamw" in n or "k_sumsq" in n:
        return "optimizer"
    if "at::native" in n or "at_cuda_detail" in n or "elementwise_kernel" in n:
        return "torch_glue"
    return "own_small"


def tree_stamp():
    """{head, source_sha16} of the tree the profile was taken on: .git_head is written next to the sources before the
    snapshot goes to the GPU box (it has no .git); the fingerprint is bench.source_fingerprint()."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    from bench import source_fingerprint
    head = None
    try:
        head = open(os.path.join(root, ".git_head")).read().strip() or None
    except OSError:
        pass
    return {"head": head, "source_sha16": source_fingerprint()}


def main():
    d = sys.argv[1]
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    top = int(sys.argv[3]) if len(sys.argv) > 3 else 30
    f = max(glob.glob(d + '/**/*kernel_trace.csv', recursive=True), key=os.path.getmtime)
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
    # the timed region: the last `steps` occurrences of the step-count kernel close a step each (one per update, whether the
    # update is one AdamW launch or two; traces older than that kernel: the AdamW launch)
    idx = [i for i, r in enumerate(rows) if 'k_final_sum_step' in r['Kernel_Name']]
    if not idx:
        idx = [i for i, r in enumerate(rows) if 'k_adamw' in r['Kernel_Name']]
    first = idx[-steps - 1] + 1
    sel = rows[first:idx[-1] + 1]
    wall = (int(sel[-1]['End_Timestamp']) - int(sel[0]['Start_Timestamp'])) / steps / 1e3
    c = collections.Counter(); t = collections.Counter()
    for r in sel:
        n = r['Kernel_Name']; c[n] += 1; t[n] += int(r['End_Timestamp']) - int(r['Start_Timestamp'])
    tot = sum(t.values()) / steps / 1e3
    # whole-device gaps above 300 us are the profiler's own stalls (its buffers being flushed: a handful per run, ~1.3 ms
    # each, in front of arbitrary kernels), not the program's: the wall per step is also given without them
    stall, nstall, be = 0, 0, int(sel[0]['Start_Timestamp'])
    for r in sel:
        s0 = int(r['Start_Timestamp'])
        if s0 - be > 300000:
            stall += s0 - be; nstall += 1
        be = max(be, int(r['End_Timestamp']))
    wall_ns = wall - stall / steps / 1e3
    print(f"kernels/step {len(sel)/steps:.0f}  wall {wall:.0f} us/step ({wall_ns:.0f} without the profiler's {nstall} stalls of > 300 us)  kernel sum {tot:.0f} us/step")
    small_n = sum(c[n] for n in c if t[n] / c[n] < 12000) / steps
    small = sum(v for n, v in t.items() if v / c[n] < 12000) / steps / 1e3
    print(f"kernels < 12 us: {small_n:.0f}/step, {small:.0f} us/step")
    fam_t = collections.Counter(); fam_c = collections.Counter()
    for n in t:
        fam_t[family(n)] += t[n]; fam_c[family(n)] += c[n]
    for k, v in sorted(fam_t.items(), key=lambda x: -x[1]):
        print(f"family {k:20s} {fam_c[k]/steps:6.1f}/step {v/steps/1e3:8.1f} us/step")
    for n, v in sorted(t.items(), key=lambda x: -x[1])[:top]:
        print(f"{c[n]/steps:6.1f}/step {v/c[n]/1e3:8.1f} us  {v/steps/1e3:7.1f} us/step  {n[:110]}")
    # device idle: intervals inside the timed region in which NO kernel of any queue runs, charged to the kernel that ends them
    idle = collections.Counter(); idle_n = collections.Counter()
    busy_end = int(sel[0]['Start_Timestamp']); idle_total = 0
    for r in sel:
        s0, e0 = int(r['Start_Timestamp']), int(r['End_Timestamp'])
        if s0 > busy_end:
            gap = s0 - busy_end
            idle_total += gap
            if gap > 3000:
                idle[r['Kernel_Name']] += gap; idle_n[r['Kernel_Name']] += 1
        busy_end = max(busy_end, e0)
    print(f"device idle (no kernel on any queue): {idle_total / steps / 1e3:.0f} us/step; gaps > 3 us by the kernel that follows:")
    for n, v in sorted(idle.items(), key=lambda x: -x[1])[:14]:
        print(f"   idle {v / steps / 1e3:6.1f} us/step  ({idle_n[n] / steps:.1f} gaps/step, {v / idle_n[n] / 1e3:5.1f} us each) before {n[:90]}")
    if len(sys.argv) > 6:      # region report: launches between two kernel-name patterns, for a few steps (argv[6] = "patA,patB")
        pa, pb = sys.argv[6].split(",")
        shown = 0
        i = first
        while i < len(rows) and shown < 4:
            if pa in rows[i]['Kernel_Name']:
                j = i
                while j < len(rows) and pb not in rows[j]['Kernel_Name']:
                    j += 1
                t0 = int(rows[i]['Start_Timestamp'])
                print(f"--- region {pa} .. {pb}, occurrence {shown}")
                for r in rows[i:j + 1]:
                    print(f"{(int(r['Start_Timestamp']) - t0) / 1e3:9.1f} us  +{(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3:7.1f} us  q{r.get('Queue_Id', '?')}  {r['Kernel_Name'][:100]}")
                shown += 1
                i = j + len(sel) // steps * 9        # skip ahead ~9 steps
            else:
                i += 1
    if len(sys.argv) > 5:      # the launches of the last timed step, in start order (stream = queue id): who sits between whom
        last = rows[idx[-2] + 1:idx[-1] + 1]
        t0 = int(last[0]['Start_Timestamp'])
        with open(sys.argv[5], "w") as fh:
            for r in last:
                fh.write(f"{(int(r['Start_Timestamp']) - t0) / 1e3:9.1f} us  +{(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3:7.1f} us  "
                         f"q{r.get('Queue_Id', '?')}  {r['Kernel_Name'][:150]}\n")
    if len(sys.argv) > 4:
        out = {**tree_stamp(), "steps": steps, "wall_us_per_step": wall, "wall_us_per_step_without_profiler_stalls": wall_ns, "kernel_sum_us_per_step": tot, "launches_per_step": len(sel) / steps,
               "under_12us": {"launches_per_step": small_n, "us_per_step": small},
               "families": {k: {"us_per_step": fam_t[k] / steps / 1e3, "launches_per_step": fam_c[k] / steps} for k in fam_t},
               "top_kernels": [{"name": n[:160], "per_step": c[n] / steps, "avg_us": v / c[n] / 1e3, "us_per_step": v / steps / 1e3}
                               for n, v in sorted(t.items(), key=lambda x: -x[1])[:12]]}
        mf = os.environ.get("SPADOT_MFMA_SUMMARY")          # tools/mfma_summary.py output of the same tree: carried along
        if mf and os.path.exists(mf):
            out["mfma"] = json.load(open(mf))
        json.dump(out, open(sys.argv[4], "w"), indent=1)


if __name__ == "__main__":
    main()
