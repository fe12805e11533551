# same-box A/B of the training step over several environment settings, interleaved twice:
#   bash tools/ab_step.sh "A=1" "A=0" ...      (each setting runs bench.py --leg train; prints steps/s)
set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for round in 1 2; do
  i=0
  for cfg in "$@"; do
    i=$((i+1))
    env $cfg timeout -k 10 400 python bench.py --leg train --no-cpu-baseline --no-epoch --repeats 7 > gpurun_out/abs_${i}_$round.json 2> gpurun_out/abs_${i}_$round.err || { tail -5 gpurun_out/abs_${i}_$round.err; exit 1; }
    python tools/bench_value.py "$cfg" < gpurun_out/abs_${i}_$round.json
  done
done
