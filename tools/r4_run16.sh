# round 4, GPU call 16: the SVGP forward hands p_m / p_v to the tail first (rest of the ELBO in the svgp_pre stage)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_model_gpu.py -x -q -k "svgp or composite or reduction" > gpurun_out/r4_t16.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r4_t16.log
tail -4 gpurun_out/r4_t16.log
bash tools/ab_step.sh "SPADOT_SVGP_ELBO_LATE=0" "SPADOT_SVGP_ELBO_LATE=1" 2>&1 | tee gpurun_out/r4_ab16.txt
timeout -k 10 300 python tools/stage_stamps.py > gpurun_out/r4_stamps16.txt 2>&1; tail -15 gpurun_out/r4_stamps16.txt
timeout -k 10 600 python -m pytest tests/test_train_gpu.py tests/test_step_parity_gpu.py -x -q -k "not full_size_inference and not cfg5 and not cfg2 and not cfg4_width" > gpurun_out/r4_t16b.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r4_t16b.log
tail -4 gpurun_out/r4_t16b.log
