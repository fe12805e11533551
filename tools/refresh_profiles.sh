#!/bin/bash
# Regenerates everything under profiles/rNN from ONE box (run through gpurun from the repo root):
#   gpurun -- 'bash tools/refresh_profiles.sh r01' && cp gpurun_out/r01/summaries/* profiles/r01/
# kernel trace + stats of both bench legs, the two PMC passes of the Sinkhorn leg (one counter per pass, as the
# guide's HBM recipe says), then the plain bench line.  Raw output goes to gpurun_out/, the summaries to profiles/.
set -eo pipefail
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O $R/profiles/$TAG
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/sink -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --leg sinkhorn > $O/sink.json 2> $O/sink.log
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --leg sinkhorn > /dev/null 2> $O/pmc_fetch.log
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --leg sinkhorn > /dev/null 2> $O/pmc_write.log
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/train -- python3 $R/bench.py --steps 40 --warmup 5 --no-cpu-baseline --leg train > $O/train.json 2> $O/train.log
cd $R
P=profiles/$TAG
python3 tools/pmc_summary.py $O/pmc_fetch $O/pmc_write $P/sinkhorn_cfg3_f32_pmc_summary.csv
cp "$(ls -t $(find $O/sink -name '*kernel_stats.csv') | head -1)" $P/sinkhorn_cfg3_f32_kernel_stats.csv
cp "$(ls -t $(find $O/train -name '*kernel_stats.csv') | head -1)" $P/train_cfg3_bf16_kernel_stats.csv
tail -1 $O/sink.json > $P/sinkhorn_cfg3_f32_bench_under_rocprof.json
tail -1 $O/train.json > $P/train_cfg3_bf16_bench_under_rocprof.json
python3 tools/prof_summary.py $O/train 40 40 > $P/train_cfg3_bf16_per_step_breakdown.txt
# the plain line (reads the PMC summary just written for roofline.traffic)
timeout -k 10 800 python3 bench.py > $O/bench.json 2> $O/bench.log
tail -1 $O/bench.json > $P/bench_cfg3_1gpu.json
# keep gpurun_out small enough to be merged back
find $O -name '*kernel_trace.csv' -size +20M -delete
find $O -name '*counter_collection.csv' -size +20M -delete
# only gpurun_out/ travels back from the box: leave a copy of the summaries there (copy it into profiles/ afterwards)
mkdir -p $O/summaries && cp $P/* $O/summaries/
python3 tools/bench_value.py final < $P/bench_cfg3_1gpu.json
