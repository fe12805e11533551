# same-box A/B of BUILD variants of csrc/gat_mfma.hip on tools/agg_variants.py:  bash tools/ab_agg_build.sh "" "-DAGG_RING=2 -DAGG_WGS=3" ...
set -e
cd $GRAFT_REPO_ROOT
for round in 1 2; do
for defs in "$@"; do
  touch spadot_amd/csrc/gat_mfma.hip
  SPADOT_BUILD_DEFS="$defs" python -m spadot_amd.csrc.build > gpurun_out/abg_build.log 2>&1 || { tail -5 gpurun_out/abg_build.log; exit 1; }
  echo "== [$defs]"
  PYTHONPATH=. timeout -k 10 200 python tools/agg_variants.py 2>/dev/null | head -4 | tr '\n' ' '; echo
done
done
touch spadot_amd/csrc/gat_mfma.hip
python -m spadot_amd.csrc.build > /dev/null 2>&1
