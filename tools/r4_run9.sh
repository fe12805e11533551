# round 4, GPU call 9: deferred weight gradients, cost setup on the matrix cores
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 700 python -m pytest tests/test_train_gpu.py tests/test_step_parity_gpu.py tests/test_gat_tail_gpu.py -x -q -k "not full_size_inference and not cfg5 and not cfg2" > gpurun_out/r4_t9.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r4_t9.log
tail -4 gpurun_out/r4_t9.log
bash tools/ab_step.sh "SPADOT_DEFER_WGRAD=0" "SPADOT_DEFER_WGRAD=1" "SPADOT_DEFER_WGRAD=1 SPADOT_SWEEP_LDS_KB=140" "SPADOT_DEFER_WGRAD=0 SPADOT_SWEEP_LDS_KB=140" 2>&1 | tee gpurun_out/r4_ab9.txt
timeout -k 10 300 python tools/stage_stamps.py > gpurun_out/r4_stamps9.txt 2>&1; tail -14 gpurun_out/r4_stamps9.txt
SPADOT_SWEEP_LDS_KB=140 timeout -k 10 300 python tools/stage_stamps.py > gpurun_out/r4_stamps9b.txt 2>&1; tail -14 gpurun_out/r4_stamps9b.txt
bash tools/prof_tl.sh r4i > gpurun_out/r4_tl9.log 2>&1; tail -2 gpurun_out/r4_tl9.log
timeout -k 10 400 python -m pytest tests/test_ot_gpu.py -x -q > gpurun_out/r4_t9ot.log 2>&1; echo "pytest-ot rc=$?" >> gpurun_out/r4_t9ot.log; tail -3 gpurun_out/r4_t9ot.log
for m in 0 1 0 1; do SPADOT_OT_COST_MFMA=$m timeout -k 10 120 python tools/cost_setup_time.py 2>/dev/null | head -1 | sed "s/^/MFMA=$m /"; done | tee gpurun_out/r4_cost_ab.txt
