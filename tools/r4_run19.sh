# round 4, GPU call 19: T at the head of the SVGP backward instead of beside the tail?; whole GPU suite; the plain bench line
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
bash tools/ab_step.sh "SPADOT_SVGP_T_LATE=0" "SPADOT_SVGP_T_LATE=1" 2>&1 | tee gpurun_out/r4_ab19.txt
SPADOT_SVGP_T_LATE=1 timeout -k 10 300 python tools/stage_stamps.py > gpurun_out/r4_stamps19.txt 2>&1; tail -15 gpurun_out/r4_stamps19.txt
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r4_t19.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r4_t19.log
tail -4 gpurun_out/r4_t19.log
timeout -k 10 900 python bench.py > gpurun_out/r4_bench19.json 2> gpurun_out/r4_bench19.err; tail -c 1500 gpurun_out/r4_bench19.json
