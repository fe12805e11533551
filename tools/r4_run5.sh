# round 4, GPU call 5: one-shot fp32 maps, SVGP head first
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gemm_gpu.py tests/test_model_gpu.py -x -q > gpurun_out/r4_t5.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r4_t5.log
tail -3 gpurun_out/r4_t5.log
SPADOT_SVGP_HEAD=1 timeout -k 10 600 python -m pytest tests/test_train_gpu.py -x -q -k "graphed_steps or chained or staged_replay or full_size" > gpurun_out/r4_t5b.log 2>&1; echo "pytest-head rc=$?" >> gpurun_out/r4_t5b.log
tail -3 gpurun_out/r4_t5b.log
bash tools/ab_step.sh "SPADOT_SGEMM_SLICES=0 SPADOT_HIDDEN_SLICES=0" "SPADOT_SGEMM_SLICES=1 SPADOT_HIDDEN_SLICES=0" "SPADOT_SGEMM_SLICES=1 SPADOT_HIDDEN_SLICES=1" "SPADOT_SVGP_HEAD=1" "SPADOT_SVGP_HEAD=1 SPADOT_SGEMM_SLICES=0 SPADOT_HIDDEN_SLICES=0" 2>&1 | tee gpurun_out/r4_ab5.txt
SPADOT_SVGP_HEAD=1 timeout -k 10 300 python tools/stage_stamps.py > gpurun_out/r4_stamps5.txt 2>&1; tail -14 gpurun_out/r4_stamps5.txt
bash tools/prof_tl.sh r4e > gpurun_out/r4_tl5.log 2>&1; tail -2 gpurun_out/r4_tl5.log
bash tools/prof_tl.sh r4f SPADOT_SVGP_HEAD=1 > gpurun_out/r4_tl5b.log 2>&1; tail -2 gpurun_out/r4_tl5b.log
