# round 4, GPU call 11: whole GPU suite on the restored tree, then side-stream priority A/B
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r4_t11.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r4_t11.log
tail -4 gpurun_out/r4_t11.log
bash tools/ab_step.sh "SPADOT_SIDE_PRIORITY=0" "SPADOT_SIDE_PRIORITY=-1" 2>&1 | tee gpurun_out/r4_ab11.txt
SPADOT_SIDE_PRIORITY=-1 timeout -k 10 300 python tools/stage_stamps.py > gpurun_out/r4_stamps11.txt 2>&1; tail -14 gpurun_out/r4_stamps11.txt
