"""Matrix-core utilisation per kernel from ONE rocprofv3 --pmc pass of `bench.py --leg train` (eager launches:
SPADOT_BENCH_NO_GRAPHS=1) that collected SQ_VALU_MFMA_BUSY_CYCLES, SQ_BUSY_CYCLES and GRBM_GUI_ACTIVE.

usage: mfma_summary.py <pass_dir> <out.json> [<out.csv>]

Reading (MI355X_MICROARCH.md): SQ_VALU_MFMA_BUSY_CYCLES counts matrix-pipe busy cycles summed over the chip's 1024 SIMDs
(32 per v_mfma_f32_32x32x16_bf16); GRBM_GUI_ACTIVE is the kernel's active clock count summed over the 8 XCDs.  So
    mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (1024 * GRBM_GUI_ACTIVE / 8)
is the fraction of all matrix-pipe cycles of the launch that issued MFMA work (1.0 = every SIMD's matrix pipe busy for the
whole launch = the dense bf16 peak at the clock the chip held)."""
import collections
import csv
import glob
import json
import os
import sys

WANT = ("k_gemm_bf16", "k_gemm_wgrad_bf16", "k_gat_agg", "k_gat_edot", "Cijk_", "Custom_Cijk")


def main():
    d, out_json = sys.argv[1], sys.argv[2]
    f = max(glob.glob(d + "/**/*counter_collection.csv", recursive=True), key=os.path.getmtime)
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if any(w in k for w in WANT):
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    rows = []
    for k, c in acc.items():
        n = len(c.get("GRBM_GUI_ACTIVE", []))
        if n == 0:
            continue
        mean = lambda name: sum(c.get(name, [0.0])) / max(1, len(c.get(name, [0.0])))
        mf, gui, sqb = mean("SQ_VALU_MFMA_BUSY_CYCLES"), mean("GRBM_GUI_ACTIVE"), mean("SQ_BUSY_CYCLES")
        if mf <= 0:
            continue
        rows.append({"kernel": k[:150], "launches": n, "SQ_VALU_MFMA_BUSY_CYCLES": mf, "GRBM_GUI_ACTIVE": gui,
                     "SQ_BUSY_CYCLES": sqb, "mfma_util": mf / (128.0 * gui) if gui > 0 else None})
    rows.sort(key=lambda r: -r["SQ_VALU_MFMA_BUSY_CYCLES"] * r["launches"])
    json.dump({"what": "per-launch means; mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (128 * GRBM_GUI_ACTIVE) "
                       "(1024 SIMDs, GUI_ACTIVE summed over 8 XCDs)", "kernels": rows}, open(out_json, "w"), indent=1)
    if len(sys.argv) > 3:
        with open(sys.argv[3], "w", newline="") as fh:
            w = csv.writer(fh)
            w.writerow(["kernel", "launches", "SQ_VALU_MFMA_BUSY_CYCLES_mean", "GRBM_GUI_ACTIVE_mean", "SQ_BUSY_CYCLES_mean", "mfma_util"])
            for r in rows:
                w.writerow([r["kernel"], r["launches"], r["SQ_VALU_MFMA_BUSY_CYCLES"], r["GRBM_GUI_ACTIVE"], r["SQ_BUSY_CYCLES"], r["mfma_util"]])
    for r in rows[:16]:
        print(f"{r['mfma_util']:.3f}  x{r['launches']:4d}  {r['kernel'][:110]}")


if __name__ == "__main__":
    main()
