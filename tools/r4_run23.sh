# round 4, GPU call 23: the SVGP encoder's first map on the matrix cores from the optimizer-kept bf16 image
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
bash tools/ab_step.sh "SPADOT_FIRST_MAP_BF16=0" "SPADOT_FIRST_MAP_BF16=1" 2>&1 | tee gpurun_out/r4_ab23.txt
timeout -k 10 300 python tools/stage_stamps.py > gpurun_out/r4_stamps23.txt 2>&1; tail -15 gpurun_out/r4_stamps23.txt
timeout -k 10 900 python -m pytest tests/test_train_gpu.py tests/test_step_parity_gpu.py tests/test_model_gpu.py -x -q -k "not full_size_inference and not cfg5 and not cfg2 and not cfg4_width" > gpurun_out/r4_t23.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r4_t23.log
tail -4 gpurun_out/r4_t23.log
