# round 4, GPU call 15: q1 through T by default; precompute stage on a normal-priority stream?; timeline
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_model_gpu.py -x -q -k "svgp or composite or reduction" > gpurun_out/r4_t15.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r4_t15.log
tail -4 gpurun_out/r4_t15.log
bash tools/ab_step.sh "SPADOT_PRE_STREAM=0" "SPADOT_PRE_STREAM=1" 2>&1 | tee gpurun_out/r4_ab15.txt
SPADOT_PRE_STREAM=1 timeout -k 10 300 python tools/stage_stamps.py > gpurun_out/r4_stamps15.txt 2>&1; tail -15 gpurun_out/r4_stamps15.txt
bash tools/prof_tl.sh r4k > gpurun_out/r4_tl15.log 2>&1; tail -2 gpurun_out/r4_tl15.log
