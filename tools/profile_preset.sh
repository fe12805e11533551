#!/bin/bash
# Kernel trace of the training leg of one bench preset -> profiles/<tag>/train_<preset>_<dtype>_{families.json,per_step_breakdown.txt,
# step_timeline.txt,kernel_stats.csv} (+ the bench line under the profiler).  Run through gpurun from the repo root:
#   gpurun -- 'bash tools/profile_preset.sh cfg2 f32 r05'       then copy gpurun_out/<tag>/summaries/* into profiles/<tag>/
set -eo pipefail
PRESET=$1; DT=$2; TAG=${3:-r05}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG/$PRESET
mkdir -p $O $R/profiles/$TAG $R/gpurun_out/$TAG/summaries
cd /tmp && export TMPDIR=/tmp
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/train -- python3 $R/bench.py --preset $PRESET --steps 40 --warmup 5 --repeats 1 --no-epoch --no-cpu-baseline --leg train > $O/train.json 2> $O/train.log
cd $R
P=profiles/$TAG
N=train_${PRESET}_${DT}
cp "$(ls -t $(find $O/train -name '*kernel_stats.csv') | head -1)" $P/${N}_kernel_stats.csv
tail -1 $O/train.json > $P/${N}_bench_under_rocprof.json
python3 tools/prof_summary.py $O/train 40 40 $P/${N}_families.json > $P/${N}_per_step_breakdown.txt
python3 tools/prof_timeline.py $O/train 0 1 > $P/${N}_step_timeline.txt
timeout -k 10 600 python3 bench.py --preset $PRESET --leg train --no-cpu-baseline > $O/bench.json 2> $O/bench.log
tail -1 $O/bench.json > $P/bench_${PRESET}_1gpu.json
find $O -name '*kernel_trace.csv' -size +20M -delete
cp $P/${N}_* $P/bench_${PRESET}_1gpu.json $R/gpurun_out/$TAG/summaries/
python3 tools/bench_value.py "$PRESET final" < $P/bench_${PRESET}_1gpu.json
head -40 $P/${N}_per_step_breakdown.txt
