# round 4, GPU call 8: whole suite (no -x), timeline + stamps of the current tree
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -m gpu -q > gpurun_out/r4_t8.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r4_t8.log
tail -6 gpurun_out/r4_t8.log
timeout -k 10 300 python tools/stage_stamps.py > gpurun_out/r4_stamps8.txt 2>&1; tail -12 gpurun_out/r4_stamps8.txt
bash tools/prof_tl.sh r4h > gpurun_out/r4_tl8.log 2>&1; tail -2 gpurun_out/r4_tl8.log
