# round 4, GPU call 2: the aggregate-first last layer -- tests, stage stamps, A/B, timeline
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gat_tail_gpu.py "tests/test_train_gpu.py::test_weight_images_follow_every_writer_of_the_weights" tests/test_step_parity_gpu.py -x -q -s > gpurun_out/r4_t2.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r4_t2.log
tail -3 gpurun_out/r4_t2.log
SPADOT_GAT_TAIL=0 timeout -k 10 300 python tools/stage_stamps.py > gpurun_out/r4_stamps_tail0.txt 2>&1; tail -12 gpurun_out/r4_stamps_tail0.txt
SPADOT_GAT_TAIL=1 timeout -k 10 300 python tools/stage_stamps.py > gpurun_out/r4_stamps_tail1.txt 2>&1; tail -12 gpurun_out/r4_stamps_tail1.txt
bash tools/ab_step.sh "SPADOT_GAT_TAIL=0" "SPADOT_GAT_TAIL=1" 2>&1 | tee gpurun_out/r4_ab_tail.txt
bash tools/prof_tl.sh r4tail SPADOT_GAT_TAIL=1 > gpurun_out/r4_tl.log 2>&1; tail -2 gpurun_out/r4_tl.log
