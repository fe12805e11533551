# same-box A/B over several environment settings:  bash tools/ab_multi.sh "A=1 B=0" "A=0 B=0" ...
set -e
cd $GRAFT_REPO_ROOT
i=0
for cfg in "$@"; do
  i=$((i+1))
  env $cfg timeout -k 10 400 python bench.py --leg train --no-cpu-baseline --no-epoch --repeats 7 > gpurun_out/abm_$i.json 2> gpurun_out/abm_$i.err
  python tools/bench_value.py "$cfg" < gpurun_out/abm_$i.json
done
