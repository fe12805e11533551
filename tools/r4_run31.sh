# round 4, GPU call 31: BatchNorm kernel geometry (columns x row lanes)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for round in 1 2; do
for d in "16 16" "16 32" "8 32" "4 64"; do
  set -- $d
  touch spadot_amd/csrc/model_kernels.hip
  SPADOT_BUILD_DEFS="-DBN_COLS_DEF=$1 -DBN_RG_DEF=$2" python -m spadot_amd.csrc.build > gpurun_out/abb_build.log 2>&1 || { tail -5 gpurun_out/abb_build.log; exit 1; }
  timeout -k 10 400 python bench.py --leg train --no-cpu-baseline --no-epoch --repeats 7 > gpurun_out/abb.json 2> gpurun_out/abb.err
  python tools/bench_value.py "BN_COLS=$1 BN_RG=$2" < gpurun_out/abb.json
done
done 2>&1 | tee gpurun_out/r4_ab31.txt
