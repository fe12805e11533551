# round 4, GPU call 14: SVGP backward restructured (dt / K dt on the mid kernels, q1 through T), colsum launches deferred
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_model_gpu.py tests/test_mlp_chain_gpu.py -x -q -k "svgp or mlp_chain or headfc or head_fc or composite" > gpurun_out/r4_t14.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r4_t14.log
tail -4 gpurun_out/r4_t14.log
bash tools/ab_step.sh "SPADOT_SVGP_MIDBWD=0" "SPADOT_SVGP_MIDBWD=1" "SPADOT_SVGP_MIDBWD=1 SPADOT_SVGP_Q1T=1" 2>&1 | tee gpurun_out/r4_ab14.txt
SPADOT_SVGP_Q1T=1 timeout -k 10 300 python tools/stage_stamps.py > gpurun_out/r4_stamps14.txt 2>&1; tail -15 gpurun_out/r4_stamps14.txt
timeout -k 10 600 python -m pytest tests/test_train_gpu.py tests/test_step_parity_gpu.py -x -q -k "not full_size_inference and not cfg5 and not cfg2 and not cfg4_width" > gpurun_out/r4_t14b.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r4_t14b.log
tail -4 gpurun_out/r4_t14b.log
