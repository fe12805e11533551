# round 4, GPU call 28: a throttled streaming update (fewer workgroups) beside the next step's SVGP encoder?
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
bash tools/ab_step.sh "SPADOT_ADAMW_MAX_WGS=4096" "SPADOT_ADAMW_MAX_WGS=1024" "SPADOT_ADAMW_MAX_WGS=512" "SPADOT_ADAMW_MAX_WGS=256" 2>&1 | tee gpurun_out/r4_ab28.txt
SPADOT_ADAMW_MAX_WGS=512 timeout -k 10 300 python tools/stage_stamps.py > gpurun_out/r4_stamps28.txt 2>&1; tail -16 gpurun_out/r4_stamps28.txt
