# per-kernel microseconds of the GAT edge kernels (tools/gat_bench.py under rocprofv3 --kernel-trace --stats) for BUILD variants of
# csrc/gat_mfma.hip:   bash tools/gat_probe.sh "" "-DEDOT_PROBE=1" ...
set -e
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
mkdir -p gpurun_out/gat_probe
for defs in "$@"; do
  touch spadot_amd/csrc/gat_mfma.hip
  SPADOT_BUILD_DEFS="$defs" python -m spadot_amd.csrc.build > gpurun_out/gat_probe/build.log 2>&1 || { tail -5 gpurun_out/gat_probe/build.log; exit 1; }
  echo "== [$defs]"
  rm -rf gpurun_out/gat_probe/tr
  (cd /tmp && export TMPDIR=/tmp && GAT_BENCH_REPS=40 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/gat_probe/tr -- python3 $R/tools/gat_bench.py > $R/gpurun_out/gat_probe/bench.txt 2> $R/gpurun_out/gat_probe/err.txt) || { tail -5 gpurun_out/gat_probe/err.txt; exit 1; }
  python3 - <<'PY'
import csv, glob, os
f = max(glob.glob(os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/gat_probe/tr/**/*kernel_stats.csv", recursive=True), key=os.path.getmtime)
for r in csv.DictReader(open(f)):
    n = r["Name"]
    if "k_gat_" in n or "colsum" in n:
        print("   %-28s calls %5s avg %8.1f us" % (n.split("(")[0].replace("void (anonymous namespace)::", "")[:28], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
done
touch spadot_amd/csrc/gat_mfma.hip
python -m spadot_amd.csrc.build > /dev/null 2>&1
