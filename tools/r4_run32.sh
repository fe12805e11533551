cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
bash tools/ab_step.sh "SPADOT_CLUSTER_FORK=0" "SPADOT_CLUSTER_FORK=1" 2>&1 | tee gpurun_out/r4_ab32.txt
SPADOT_CLUSTER_FORK=1 timeout -k 10 600 python -m pytest tests/test_train_gpu.py tests/test_step_parity_gpu.py -x -q -k "not full_size_inference and not cfg5 and not cfg2 and not cfg4_width" 2>&1 | tail -3
