# round 4, GPU call 10: loss values off the chain
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 700 python -m pytest tests/test_train_gpu.py tests/test_step_parity_gpu.py tests/test_model_gpu.py -x -q -k "not full_size_inference and not cfg5 and not cfg2 and not cfg4_width" > gpurun_out/r4_t10.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r4_t10.log
tail -4 gpurun_out/r4_t10.log
bash tools/ab_step.sh "SPADOT_DEFER_WGRAD=0" "SPADOT_DEFER_WGRAD=1" 2>&1 | tee gpurun_out/r4_ab10.txt
timeout -k 10 300 python tools/stage_stamps.py > gpurun_out/r4_stamps10.txt 2>&1; tail -14 gpurun_out/r4_stamps10.txt
bash tools/prof_tl.sh r4j > gpurun_out/r4_tl10.log 2>&1; tail -2 gpurun_out/r4_tl10.log
