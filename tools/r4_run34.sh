# round 4, GPU call 34: k_gat_edot with two heads per workgroup (A/B builds)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for round in 1 2; do
for d in "" "-DEDOT_HPW=2 -DEDOT_PAD=0 -DEDOT_WGS=2"; do
  touch spadot_amd/csrc/gat_mfma.hip
  SPADOT_BUILD_DEFS="$d" python -m spadot_amd.csrc.build > gpurun_out/abb_build.log 2>&1 || { tail -5 gpurun_out/abb_build.log; exit 1; }
  PYTHONPATH=. timeout -k 10 200 python tools/gat_bench.py 2>/dev/null | grep -i -A1 "matrix-core" | tail -1
  timeout -k 10 400 python bench.py --leg train --no-cpu-baseline --no-epoch --repeats 7 > gpurun_out/abb.json 2> gpurun_out/abb.err
  python tools/bench_value.py "[$d]" < gpurun_out/abb.json
done
done 2>&1 | tee gpurun_out/r4_ab34.txt
timeout -k 10 300 python -m pytest tests/test_gat_mfma_gpu.py -x -q 2>&1 | tail -3
