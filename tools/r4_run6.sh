# round 4, GPU call 6: SVGP head first
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_gemm_gpu.py -x -q -k "sgemm_nt or dgemm" > gpurun_out/r4_t6.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r4_t6.log
tail -2 gpurun_out/r4_t6.log
SPADOT_SVGP_HEAD=1 timeout -k 10 600 python -m pytest tests/test_train_gpu.py tests/test_step_parity_gpu.py -x -q -k "graphed_steps or chained or staged_replay or full_size_step or full_size_chained or cfg3_training_step" > gpurun_out/r4_t6b.log 2>&1; echo "pytest-head rc=$?" >> gpurun_out/r4_t6b.log
tail -3 gpurun_out/r4_t6b.log
bash tools/ab_step.sh "SPADOT_SVGP_HEAD=0" "SPADOT_SVGP_HEAD=1" 2>&1 | tee gpurun_out/r4_ab6.txt
SPADOT_SVGP_HEAD=1 timeout -k 10 300 python tools/stage_stamps.py > gpurun_out/r4_stamps6.txt 2>&1; tail -14 gpurun_out/r4_stamps6.txt
bash tools/prof_tl.sh r4g SPADOT_SVGP_HEAD=1 > gpurun_out/r4_tl6.log 2>&1; tail -2 gpurun_out/r4_tl6.log
