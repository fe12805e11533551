# same-box A/B of the training step under one environment switch:  bash tools/ab_env.sh VAR a b a b
set -e
cd $GRAFT_REPO_ROOT
var=$1; shift
i=0
for v in "$@"; do
  i=$((i+1))
  env $var=$v timeout -k 10 400 python bench.py --leg train --no-cpu-baseline --no-epoch --repeats 7 > gpurun_out/ab_${var}_${v}_$i.json 2> gpurun_out/ab_${var}_${v}_$i.err
  python tools/bench_value.py ${var}=${v} < gpurun_out/ab_${var}_${v}_$i.json
done
