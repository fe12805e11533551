# round 4, GPU call 36: what the five-graph form (no queue: the form data-parallel ranks replay, minus the exchange) gains from
# the precomputed SVGP backward + fused cluster launch
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
bash tools/ab_step.sh "SPADOT_DEFER_WGRAD=0 SPADOT_SVGP_PRE=0 SPADOT_CLUSTER_FB=0" "SPADOT_DEFER_WGRAD=0" "SPADOT_DEFER_WGRAD=1" 2>&1 | tee gpurun_out/r4_ab36.txt
