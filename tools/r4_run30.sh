# round 4, GPU call 30: BatchNorm kernels on 256 / 128 / 64 threads (same-box A/B builds)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for round in 1 2; do
for d in 16 8 4; do
  touch spadot_amd/csrc/model_kernels.hip
  SPADOT_BUILD_DEFS="-DBN_RG_DEF=$d" python -m spadot_amd.csrc.build > gpurun_out/abb_build_$d.log 2>&1 || { tail -5 gpurun_out/abb_build_$d.log; exit 1; }
  timeout -k 10 400 python bench.py --leg train --no-cpu-baseline --no-epoch --repeats 7 > gpurun_out/abb_$d.json 2> gpurun_out/abb_$d.err
  python tools/bench_value.py "BN_RG=$d" < gpurun_out/abb_$d.json
done
done 2>&1 | tee gpurun_out/r4_ab30.txt
touch spadot_amd/csrc/model_kernels.hip
SPADOT_BUILD_DEFS="-DBN_RG_DEF=4" python -m spadot_amd.csrc.build > /dev/null 2>&1
timeout -k 10 300 python -m pytest tests/test_model_gpu.py -x -q -k "bn_act or composite" 2>&1 | tail -3
