# one rocprofv3 kernel trace of the training leg + the per-step breakdown:  bash tools/prof_train.sh NAME [env assignments...]
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
name=$1; shift
for kv in "$@"; do export "$kv"; done
O=$R/gpurun_out/prof_$name
mkdir -p $O
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/train -- python3 $R/bench.py --steps 40 --warmup 5 --repeats 1 --no-epoch --no-cpu-baseline --leg train > $O/train.json 2> $O/train.log
cd $R
python3 tools/prof_summary.py $O/train 40 40 $O/families.json $O/sequence.txt ${REGION:-k_mix_losses_fwd,k_sqerr_bwd} > $O/breakdown.txt
find $O/train -name "*.csv" ! -name "*stats*" -delete 2>/dev/null || true
