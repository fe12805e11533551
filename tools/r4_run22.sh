# round 4, GPU call 22: look-ahead sweep
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_model_gpu.py -x -q -k "sweep or spd or svgp" > gpurun_out/r4_t22.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r4_t22.log
tail -4 gpurun_out/r4_t22.log
timeout -k 10 200 python tools/spd_la_bench.py 2>&1 | tail -6
bash tools/ab_step.sh "SPADOT_SWEEP_LA=0" "SPADOT_SWEEP_LA=1" 2>&1 | tee gpurun_out/r4_ab22.txt
timeout -k 10 300 python tools/stage_stamps.py > gpurun_out/r4_stamps22.txt 2>&1; tail -15 gpurun_out/r4_stamps22.txt
