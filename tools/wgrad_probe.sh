# What bounds csrc/gemm_wgrad_bf16.hip: the kernel rebuilt without its MFMAs / LDS fragment reads / LDS-DMA (timing probes, wrong
# results) and timed alone at the layer shapes:   bash tools/wgrad_probe.sh     (on the GPU box; restores the normal build)
set -e
cd $GRAFT_REPO_ROOT
export PYTHONPATH=$PWD
for defs in "" "-DWGRAD_PROBE_NO_MFMA" "-DWGRAD_PROBE_NO_READS" "-DWGRAD_PROBE_NO_DMA" "-DWGRAD_PROBE_NO_MFMA -DWGRAD_PROBE_NO_READS" "-DWGRAD_PROBE_NO_READS -DWGRAD_PROBE_NO_DMA" "-DWGRAD_PROBE_NO_MFMA -DWGRAD_PROBE_NO_DMA"; do
  touch spadot_amd/csrc/gemm_wgrad_bf16.hip
  SPADOT_BUILD_DEFS="$defs" python -m spadot_amd.csrc.build > gpurun_out/wgp_build.log 2>&1
  echo "== [$defs]"
  timeout -k 10 200 python tools/gemm_bench.py wgrad 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        r = json.loads(l)
        if r['M'] > 1000: print('   ', r['M'], r['N'], r['K'], 'slices', r['slices'], 'tile_k', r['tile_k'], 'us', r['us'], 'library', r['us_library'])
"
done
touch spadot_amd/csrc/gemm_wgrad_bf16.hip
python -m spadot_amd.csrc.build > gpurun_out/wgp_build.log 2>&1
