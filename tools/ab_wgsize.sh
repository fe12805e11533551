set -e
cd $GRAFT_REPO_ROOT
run() { timeout -k 10 400 python bench.py --leg train --no-cpu-baseline --no-epoch --repeats 7 > gpurun_out/ab_$1.json 2> gpurun_out/ab_$1.err; python tools/bench_value.py $1 < gpurun_out/ab_$1.json; }
run cur_a
# variant 1: colsum 16 groups (1024 threads) + post_fwd 1024 threads
sed -i 's/constexpr int CS_GY = 4,/constexpr int CS_GY = 16,/; s/hipLaunchKernelGGL(k_svgp_post_fwd, dim3(1), dim3(256)/hipLaunchKernelGGL(k_svgp_post_fwd, dim3(1), dim3(1024)/' spadot_amd/csrc/model_kernels.hip
python -m spadot_amd.csrc.build > /dev/null 2>&1
run colsum1024_post1024
# variant 2: additionally BN back to 64 row lanes
sed -i 's/constexpr int BN_COLS = 16, BN_RG = 16,/constexpr int BN_COLS = 16, BN_RG = 64,/' spadot_amd/csrc/model_kernels.hip
python -m spadot_amd.csrc.build > /dev/null 2>&1
run all1024
# back to current
sed -i 's/constexpr int CS_GY = 16,/constexpr int CS_GY = 4,/; s/hipLaunchKernelGGL(k_svgp_post_fwd, dim3(1), dim3(1024)/hipLaunchKernelGGL(k_svgp_post_fwd, dim3(1), dim3(256)/; s/constexpr int BN_COLS = 16, BN_RG = 64,/constexpr int BN_COLS = 16, BN_RG = 16,/' spadot_amd/csrc/model_kernels.hip
python -m spadot_amd.csrc.build > /dev/null 2>&1
run cur_b
