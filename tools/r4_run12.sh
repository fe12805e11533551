# round 4, GPU call 12: balance of the backward pair -- where the deferred gradient work runs
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
bash tools/ab_step.sh "SPADOT_DEFER_L2=1" "SPADOT_DEFER_L2=0" "SPADOT_LATE_STREAM=1 SPADOT_DEFER_L2=1" "SPADOT_LATE_STREAM=1 SPADOT_DEFER_L2=0" "SPADOT_LATE_STREAM=1 SPADOT_LATE_PRIORITY=-1" 2>&1 | tee gpurun_out/r4_ab12.txt
SPADOT_LATE_STREAM=1 timeout -k 10 300 python tools/stage_stamps.py > gpurun_out/r4_stamps12.txt 2>&1; tail -14 gpurun_out/r4_stamps12.txt
timeout -k 10 600 python -m pytest tests/test_train_gpu.py tests/test_step_parity_gpu.py -x -q -k "not full_size_inference and not cfg5 and not cfg2 and not cfg4_width" > gpurun_out/r4_t12.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r4_t12.log
tail -4 gpurun_out/r4_t12.log
