cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
bash tools/ab_step.sh "SPADOT_ADAMW_MAX_WGS=4096" "SPADOT_ADAMW_MAX_WGS=512" "SPADOT_ADAMW_MAX_WGS=384" "SPADOT_ADAMW_MAX_WGS=768" 2>&1 | tee gpurun_out/r4_ab29.txt
