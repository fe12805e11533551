# round 4, GPU call 20: the loss tail on the side stream, the svgp_pre GEMMs on the main stream?
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
bash tools/ab_step.sh "SPADOT_TAIL_ON_SIDE=0" "SPADOT_TAIL_ON_SIDE=1" 2>&1 | tee gpurun_out/r4_ab20.txt
SPADOT_TAIL_ON_SIDE=1 timeout -k 10 300 python tools/stage_stamps.py > gpurun_out/r4_stamps20.txt 2>&1; tail -15 gpurun_out/r4_stamps20.txt
SPADOT_TAIL_ON_SIDE=1 timeout -k 10 600 python -m pytest tests/test_train_gpu.py tests/test_step_parity_gpu.py -x -q -k "not full_size_inference and not cfg5 and not cfg2 and not cfg4_width" > gpurun_out/r4_t20.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r4_t20.log
tail -4 gpurun_out/r4_t20.log
