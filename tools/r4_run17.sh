# round 4, GPU call 17: no cast launch in front of the decoder's output map; colsum launches queued
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_model_gpu.py tests/test_mlp_chain_gpu.py -x -q -k "svgp or composite or reduction or mlp_chain or adamw or recon or headfc or head_fc" > gpurun_out/r4_t17.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r4_t17.log
tail -4 gpurun_out/r4_t17.log
bash tools/ab_step.sh "SPADOT_DEC_NOCAST=0" "SPADOT_DEC_NOCAST=1" 2>&1 | tee gpurun_out/r4_ab17.txt
timeout -k 10 300 python tools/stage_stamps.py > gpurun_out/r4_stamps17.txt 2>&1; tail -15 gpurun_out/r4_stamps17.txt
timeout -k 10 900 python -m pytest tests/test_train_gpu.py tests/test_step_parity_gpu.py -x -q -k "not full_size_inference and not cfg5 and not cfg2 and not cfg4_width" > gpurun_out/r4_t17b.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r4_t17b.log
tail -4 gpurun_out/r4_t17b.log
