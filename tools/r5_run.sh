# one-off GPU call of round 5 (removed at the end of the round; results go to profiles/r05/)
set -eo pipefail
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5
mkdir -p $O
case "$1" in
tests)
  timeout -k 10 1000 python -m pytest tests -m gpu -x -q 2>&1 | tail -25 > $O/tests.txt; cat $O/tests.txt ;;
tests2)
  timeout -k 10 1100 python -m pytest tests -m gpu -x -q 2>&1 | tail -25 > $O/tests2.txt; cat $O/tests2.txt ;;
ot)
  timeout -k 10 600 python -m pytest tests/test_ot_gpu.py -x -q 2>&1 | tail -8
  timeout -k 10 300 python bench.py --leg sinkhorn --no-cpu-baseline > $O/bench_sink.json 2> $O/bench_sink.err || tail -5 $O/bench_sink.err
  python - <<'PY'
import json
d = json.loads(open("gpurun_out/r5/bench_sink.json").read().strip().splitlines()[-1])
print({k: d["sinkhorn"].get(k) for k in ("value", "cost_setup_s", "pair_end_to_end_s", "full_solve_s", "iters_per_s_with_convergence_checks")})
PY
  ;;
prof3)
  bash tools/profile_preset.sh cfg3 bf16 r05 2>&1 | tail -5
  timeout -k 10 300 python bench.py --spots 1650 --genes 2954 --timepoints 4 --leg train --no-cpu-baseline --steps 12 --warmup 2 > $O/bench_chickenheart_shape.json 2> $O/bench_ch.err || tail -5 $O/bench_ch.err
  python - <<'PY'
import json
d = json.loads(open("gpurun_out/r5/bench_chickenheart_shape.json").read().strip().splitlines()[-1])
print("chickenheart-like (4 x 1650 x 2954):", d["value"], "steps/s;", d.get("epoch"))
PY
  ;;
ablib)
  # same-box A/B of two BUILDS of libspadot_model.so (tmp_ab/libspadot_model_prev.so = the library of an earlier commit, same ABI)
  L=spadot_amd/csrc/libspadot_model.so
  cp $L /tmp/new.so
  for round in 1 2; do
    cp tmp_ab/libspadot_model_prev.so $L
    timeout -k 10 400 python bench.py --leg train --no-cpu-baseline --no-epoch --repeats 7 > $O/ablib_prev_$round.json 2> $O/ablib.err || { tail -5 $O/ablib.err; cp /tmp/new.so $L; exit 1; }
    python tools/bench_value.py "previous library" < $O/ablib_prev_$round.json
    cp /tmp/new.so $L
    timeout -k 10 400 python bench.py --leg train --no-cpu-baseline --no-epoch --repeats 7 > $O/ablib_new_$round.json 2> $O/ablib.err || { tail -5 $O/ablib.err; exit 1; }
    python tools/bench_value.py "this tree" < $O/ablib_new_$round.json
  done
  SPADOT_STAMPS=1 timeout -k 10 300 python tools/stage_stamps.py > $O/stage_stamps_unpred.txt 2> $O/stamps.err || tail -5 $O/stamps.err
  head -14 $O/stage_stamps_unpred.txt | tail -11 ;;
abfm)
  timeout -k 10 300 python -m pytest tests/test_gemm_gpu.py tests/test_model_gpu.py -x -q -k "wgrad or first_map" 2>&1 | tail -2
  bash tools/ab_step.sh "SPADOT_FM_SLICES=8" "SPADOT_FM_SLICES=2" "SPADOT_FM_SLICES=1" 2>&1 | tee $O/ab_first_map_slices.txt
  for q in 1; do
    SPADOT_FM_SLICES=$q SPADOT_STAMPS=1 timeout -k 10 300 python tools/stage_stamps.py > $O/stage_stamps_fm_$q.txt 2> $O/stamps.err || tail -5 $O/stamps.err
    echo "== $q"; head -14 $O/stage_stamps_fm_$q.txt | tail -9
  done ;;
ablate)
  bash tools/ab_step.sh "SPADOT_LATE_ORDER=heavy" "SPADOT_LATE_ORDER=fifo" 2>&1 | tee $O/ab_late_order.txt
  for q in heavy fifo; do
    SPADOT_LATE_ORDER=$q SPADOT_STAMPS=1 timeout -k 10 300 python tools/stage_stamps.py > $O/stage_stamps_late_$q.txt 2> $O/stamps.err || tail -5 $O/stamps.err
    echo "== $q"; head -14 $O/stage_stamps_late_$q.txt | tail -9
  done ;;
abq)
  timeout -k 10 300 python -m pytest tests/test_mlp_chain_gpu.py -x -q -k "recon" 2>&1 | tail -2
  bash tools/ab_step.sh "SPADOT_WGRAD_Q=late" "SPADOT_WGRAD_Q=post" "SPADOT_WGRAD_Q=inline" 2>&1 | tee $O/ab_wgrad_queue.txt
  for q in post inline; do
    SPADOT_WGRAD_Q=$q SPADOT_STAMPS=1 timeout -k 10 300 python tools/stage_stamps.py > $O/stage_stamps_wgrad_$q.txt 2> $O/stamps.err || tail -5 $O/stamps.err
    echo "== $q"; head -14 $O/stage_stamps_wgrad_$q.txt | tail -9
  done ;;
t7)
  timeout -k 10 900 python -m pytest tests/test_gat_tail_gpu.py tests/test_model_gpu.py tests/test_train_gpu.py tests/test_step_parity_gpu.py -x -q -k "tail or latent or head or glue or staged or deferred or cfg3" > $O/t7.txt 2>&1 || { grep -B2 -A14 "^>" $O/t7.txt | head -60; tail -5 $O/t7.txt; exit 1; }
  tail -3 $O/t7.txt ;;
t6)
  timeout -k 10 900 python -m pytest tests/test_gemm_gpu.py tests/test_model_gpu.py tests/test_gat_mfma_gpu.py tests/test_train_gpu.py -x -q -k "wgrad or colsum or gat or staged or deferred or cluster" > $O/t6.txt 2>&1 || { grep -B2 -A14 "^>" $O/t6.txt | head -60; tail -5 $O/t6.txt; exit 1; }
  tail -3 $O/t6.txt ;;
t5)
  timeout -k 10 900 python -m pytest tests/test_model_gpu.py tests/test_train_gpu.py tests/test_step_parity_gpu.py -x -q -k "cluster or loss or glue or staged or deferred or cfg3 or fused" > $O/t5.txt 2>&1 || { grep -B2 -A14 "^>" $O/t5.txt | head -60; tail -5 $O/t5.txt; exit 1; }
  tail -3 $O/t5.txt ;;
t4)
  timeout -k 10 900 python -m pytest tests/test_gat_mfma_gpu.py tests/test_model_gpu.py tests/test_train_gpu.py tests/test_step_parity_gpu.py -x -q -k "gat or fused or encoder or staged or deferred or cfg3" > $O/t4.txt 2>&1 || { grep -B2 -A14 "^>" $O/t4.txt | head -60; tail -5 $O/t4.txt; exit 1; }
  tail -3 $O/t4.txt
  bash tools/ab_step.sh "X=1" 2>&1 | tail -2 ;;
t3)
  timeout -k 10 900 python -m pytest tests/test_mlp_chain_gpu.py tests/test_train_gpu.py tests/test_step_parity_gpu.py -x -q -k "recon or staged or deferred or cfg3 or chained or small_timepoint" > $O/t3.txt 2>&1 || { grep -B2 -A14 "^>" $O/t3.txt | head -60; tail -5 $O/t3.txt; exit 1; }
  tail -3 $O/t3.txt
  bash tools/ab_step.sh "X=1" 2>&1 | tail -2 ;;
ab3)
  bash tools/ab_step.sh "SPADOT_RECON_FB=1" "SPADOT_RECON_FB=0" 2>&1 | tee $O/ab_recon_fb.txt
  SPADOT_STAMPS=1 timeout -k 10 300 python tools/stage_stamps.py > $O/stage_stamps_recon_fb.txt 2> $O/stamps.err || tail -5 $O/stamps.err
  head -14 $O/stage_stamps_recon_fb.txt | tail -11 ;;
ab2)
  bash tools/ab_step.sh "SPADOT_ENC_FUSED=1 SPADOT_PREMASK=1" "SPADOT_ENC_FUSED=0 SPADOT_PREMASK=1" "SPADOT_ENC_FUSED=1 SPADOT_PREMASK=0" 2>&1 | tee $O/ab_enc_premask.txt ;;
m2)
  bash tools/ab_step.sh "X=1" 2>&1 | tee $O/bench_train_now.txt
  SPADOT_STAMPS=1 timeout -k 10 300 python tools/stage_stamps.py > $O/stage_stamps_now.txt 2> $O/stamps.err || tail -5 $O/stamps.err
  cat $O/stage_stamps_now.txt ;;
prof)
  bash tools/profile_preset.sh cfg2 f32 r05 2>&1 | tail -60
  bash tools/profile_preset.sh cfg5shape bf16 r05 2>&1 | tail -60
  timeout -k 10 400 python tools/e2e_chickenheart.py > $O/e2e_chickenheart.txt 2> $O/e2e.err || tail -5 $O/e2e.err
  grep -v "^Epoch\|^Calculating\|^The graph\|^OT iter" $O/e2e_chickenheart.txt | tail -40 ;;
probe)
  bash tools/gat_probe.sh "" "-DEDOT_PROBE=1" "-DEDOT_PROBE=2" "-DEDOT_PROBE=4" "-DEDOT_PROBE=8" "-DEDOT_PROBE=7" 2>&1 | tee $O/gat_probe.txt
  timeout -k 10 300 python tools/e2e_chickenheart.py bf16 > $O/e2e_chickenheart_bf16.txt 2> $O/e2e_bf16.err || tail -4 $O/e2e_bf16.err ;;
m1)
  timeout -k 10 400 python -m pytest tests/test_gat_mfma_gpu.py tests/test_gat_tail_gpu.py tests/test_train_gpu.py -x -q -k "gat or deferred or staged or bucketed" 2>&1 | tail -5
  timeout -k 10 300 python tools/tail_f32_diag.py 2>&1 | grep -v Warn | tee $O/tail_f32_diag.txt
  bash tools/ab_step.sh "SPADOT_ATT_FLAT=0" "SPADOT_ATT_FLAT=1" 2>&1 | tee $O/ab_att_flat.txt
  SPADOT_STAMPS=1 timeout -k 10 300 python tools/stage_stamps.py > $O/stage_stamps_att_flat.txt 2> $O/stamps.err || tail -5 $O/stamps.err
  cat $O/stage_stamps_att_flat.txt
  timeout -k 10 300 python bench.py --preset cfg2 --leg train --no-cpu-baseline --repeats 5 > $O/bench_cfg2.json 2> $O/bench_cfg2.err || tail -5 $O/bench_cfg2.err
  python tools/bench_value.py cfg2 < $O/bench_cfg2.json
  timeout -k 10 400 python bench.py --preset cfg5shape --leg train --no-cpu-baseline --repeats 5 > $O/bench_cfg5shape.json 2> $O/bench_cfg5shape.err || tail -5 $O/bench_cfg5shape.err
  python tools/bench_value.py cfg5shape < $O/bench_cfg5shape.json
  timeout -k 10 500 python tools/e2e_chickenheart.py > $O/e2e_chickenheart.txt 2> $O/e2e.err || tail -5 $O/e2e.err
  grep -v "^Epoch\|^Calculating\|^The graph" $O/e2e_chickenheart.txt | tail -40 ;;
esac
