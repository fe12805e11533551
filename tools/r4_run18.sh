# round 4, GPU call 18: fused cluster losses + donation; who gets dispatched first in the tail window
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -c "import torch; print('priority range', torch.cuda.Stream(priority=5).priority, torch.cuda.Stream(priority=-5).priority)" 2>&1 | tail -1
timeout -k 10 600 python -m pytest tests/test_model_gpu.py tests/test_mlp_chain_gpu.py -x -q -k "svgp or composite or cluster or chain or recon" > gpurun_out/r4_t18.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r4_t18.log
tail -4 gpurun_out/r4_t18.log
bash tools/ab_step.sh "SPADOT_CLUSTER_FB=0" "SPADOT_CLUSTER_FB=1" "SPADOT_PRE_STREAM=1 SPADOT_PRE_PRIORITY=1" "SPADOT_MAIN_PRIORITY=-1 SPADOT_PRE_STREAM=1" "SPADOT_MAIN_PRIORITY=-1" 2>&1 | tee gpurun_out/r4_ab18.txt
SPADOT_MAIN_PRIORITY=-1 SPADOT_PRE_STREAM=1 timeout -k 10 300 python tools/stage_stamps.py > gpurun_out/r4_stamps18.txt 2>&1; tail -15 gpurun_out/r4_stamps18.txt
timeout -k 10 900 python -m pytest tests/test_train_gpu.py tests/test_step_parity_gpu.py -x -q -k "not full_size_inference and not cfg5 and not cfg2 and not cfg4_width" > gpurun_out/r4_t18b.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r4_t18b.log
tail -4 gpurun_out/r4_t18b.log
