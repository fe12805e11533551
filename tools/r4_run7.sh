# round 4, GPU call 7: whole suite with the new tests, MFMA f64 chain probe, interim profile refresh
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > gpurun_out/r4_t7.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r4_t7.log
tail -4 gpurun_out/r4_t7.log
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -o /tmp/probe tools/mfma_f64_chain_probe.hip 2>/dev/null && timeout -k 10 60 /tmp/probe | tee gpurun_out/r4_mfma_f64_chain.txt
bash tools/refresh_profiles.sh r04 > gpurun_out/r4_refresh.log 2>&1; echo "refresh rc=$?"; tail -3 gpurun_out/r4_refresh.log
