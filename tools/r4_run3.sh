# round 4, GPU call 3: whole GPU suite on the new kernels, A/B of the three switches, timeline + stamps of the default path
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r4_t3.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r4_t3.log
tail -4 gpurun_out/r4_t3.log
bash tools/ab_step.sh "SPADOT_GAT_TAIL=0 SPADOT_DGEMM_SMALL=0 SPADOT_SGEMM_SLICES=0" "SPADOT_GAT_TAIL=1 SPADOT_DGEMM_SMALL=0 SPADOT_SGEMM_SLICES=0" "SPADOT_GAT_TAIL=1 SPADOT_DGEMM_SMALL=1 SPADOT_SGEMM_SLICES=0" "SPADOT_GAT_TAIL=1 SPADOT_DGEMM_SMALL=1 SPADOT_SGEMM_SLICES=1" "SPADOT_GAT_TAIL=1 SPADOT_DGEMM_SMALL=1 SPADOT_SGEMM_SLICES=1 SPADOT_HIDDEN_SLICES=1" 2>&1 | tee gpurun_out/r4_ab3.txt
timeout -k 10 300 python tools/stage_stamps.py > gpurun_out/r4_stamps3.txt 2>&1; tail -11 gpurun_out/r4_stamps3.txt
bash tools/prof_tl.sh r4c > gpurun_out/r4_tl3.log 2>&1; tail -2 gpurun_out/r4_tl3.log
