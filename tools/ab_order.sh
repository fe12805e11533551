set -e
cd $GRAFT_REPO_ROOT
for o in mm sm ms ss mm; do
  SPADOT_ISSUE_ORDER=$o timeout -k 10 400 python bench.py --leg train --no-cpu-baseline --no-epoch --repeats 7 > gpurun_out/ab_order_$o.json 2> gpurun_out/ab_order_$o.err
  python tools/bench_value.py order_$o < gpurun_out/ab_order_$o.json
done
