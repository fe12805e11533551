# same-box A/B of BUILD variants of one translation unit:  bash tools/ab_build.sh gat_mfma.hip "" "-DAGG_RING=4" ""
# for each definition set: rebuild that unit, run the GAT microbenchmark and the training leg
set -e
cd $GRAFT_REPO_ROOT
unit=$1; shift
i=0
for defs in "$@"; do
  i=$((i+1))
  touch spadot_amd/csrc/$unit
  SPADOT_BUILD_DEFS="$defs" python -m spadot_amd.csrc.build > gpurun_out/abb_build_$i.log 2>&1
  echo "== [$defs]"
  PYTHONPATH=. timeout -k 10 200 python tools/gat_bench.py 2>/dev/null | grep -A1 "matrix-core" | tail -1
  timeout -k 10 400 python bench.py --leg train --no-cpu-baseline --no-epoch --repeats 7 > gpurun_out/abb_$i.json 2> gpurun_out/abb_$i.err
  python tools/bench_value.py "[$defs]" < gpurun_out/abb_$i.json
done
