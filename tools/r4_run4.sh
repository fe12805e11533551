# round 4, GPU call 4
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_model_gpu.py tests/test_gat_tail_gpu.py -x -q > gpurun_out/r4_t4.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r4_t4.log
tail -3 gpurun_out/r4_t4.log
bash tools/ab_step.sh "SPADOT_GAT_TAIL=0" "SPADOT_GAT_TAIL=1" "SPADOT_GEMM_DGRAD=0" "SPADOT_WGRAD_OWN=0" "SPADOT_GEMM_FWD_SHAPES=none" 2>&1 | tee gpurun_out/r4_ab4.txt
timeout -k 10 300 python tools/stage_stamps.py > gpurun_out/r4_stamps4.txt 2>&1; tail -11 gpurun_out/r4_stamps4.txt
bash tools/prof_tl.sh r4d > gpurun_out/r4_tl4.log 2>&1; tail -2 gpurun_out/r4_tl4.log
EPOCHS=30 timeout -k 10 900 python tools/quality_run_dp.py > gpurun_out/r4_quality_dp.txt 2>&1; tail -12 gpurun_out/r4_quality_dp.txt
