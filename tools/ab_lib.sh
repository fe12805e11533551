#!/bin/bash
# Same-box A/B of two BUILDS of libspadot_model.so (same ABI): bench.py --leg train with an earlier commit's library and with the
# tree's, interleaved twice, then the stage stamps of the tree's.  Prepare the earlier library here (no GPU needed), e.g.
#   git archive <commit> spadot_amd/csrc include | tar -x -C /tmp/prev && (cd /tmp/prev/spadot_amd/csrc && python build.py)
#   mkdir -p tmp_ab && cp /tmp/prev/spadot_amd/csrc/libspadot_model.so tmp_ab/libspadot_model_prev.so     (*.so is git-ignored)
# and run through gpurun from the repo root:  gpurun -- 'bash tools/ab_lib.sh'
set -eo pipefail
cd ${GRAFT_REPO_ROOT:-$(pwd)}
O=gpurun_out/ab_lib
mkdir -p $O
L=spadot_amd/csrc/libspadot_model.so
cp $L /tmp/new.so
trap 'cp /tmp/new.so $L' EXIT
for round in 1 2; do
  cp tmp_ab/libspadot_model_prev.so $L
  timeout -k 10 400 python bench.py --leg train --no-cpu-baseline --no-epoch --repeats 7 > $O/prev_$round.json 2> $O/err.txt || { tail -5 $O/err.txt; exit 1; }
  python tools/bench_value.py "previous library" < $O/prev_$round.json
  cp /tmp/new.so $L
  timeout -k 10 400 python bench.py --leg train --no-cpu-baseline --no-epoch --repeats 7 > $O/new_$round.json 2> $O/err.txt || { tail -5 $O/err.txt; exit 1; }
  python tools/bench_value.py "this tree" < $O/new_$round.json
done
SPADOT_STAMPS=1 timeout -k 10 300 python tools/stage_stamps.py > $O/stage_stamps.txt 2> $O/err.txt || tail -5 $O/err.txt
head -14 $O/stage_stamps.txt | tail -11
