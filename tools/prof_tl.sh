# kernel-trace timeline (queue ids, start, duration) of the last profiled training step:  bash tools/prof_tl.sh NAME [ENV=VAL ...]
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
name=$1; shift
for kv in "$@"; do export "$kv"; done
O=$R/gpurun_out/tl_$name
mkdir -p $O
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $O/train -- python3 $R/bench.py --steps 20 --warmup 5 --repeats 1 --no-epoch --no-cpu-baseline --leg train > $O/train.json 2> $O/train.log
cd $R
python3 tools/prof_timeline.py $O/train 0 > $O/timeline.txt
python3 tools/prof_timeline.py $O/train 0 ${STEP_BACK:-3} > $O/timeline_b.txt || true
find $O/train -name "*.csv" -delete
tail -1 $O/timeline.txt
