#!/bin/bash
# Regenerates everything under profiles/rNN from ONE box (run through gpurun from the repo root):
#   git rev-parse HEAD > .git_head; gpurun -- 'bash tools/refresh_profiles.sh r03' && cp gpurun_out/r03/summaries/* profiles/r03/
# kernel trace + stats of both bench legs, the PMC passes of the Sinkhorn leg and of the GAT edge kernels (one counter
# group per pass, nothing but the kernel trace beside them, as the guide's HBM recipe says), then the plain bench line.
# Raw output goes to gpurun_out/, the summaries to profiles/.
set -eo pipefail
TAG=${1:-r03}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O $R/profiles/$TAG
cd /tmp && export TMPDIR=/tmp
# the Sinkhorn leg's profile runs without speculative batches: every k_fused_pass launch then sweeps the matrix (a launch
# enqueued behind a converged batch returns on the stop word in ~5 us and would pull the per-launch means down)
export SPADOT_OT_SPEC_BATCHES=1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/sink -- python3 $R/bench.py --steps 20 --warmup 3 --repeats 1 --no-cpu-baseline --leg sinkhorn > $O/sink.json 2> $O/sink.log
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python3 $R/bench.py --steps 2 --warmup 1 --repeats 1 --no-cpu-baseline --leg sinkhorn > /dev/null 2> $O/pmc_fetch.log
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python3 $R/bench.py --steps 2 --warmup 1 --repeats 1 --no-cpu-baseline --leg sinkhorn > /dev/null 2> $O/pmc_write.log
unset SPADOT_OT_SPEC_BATCHES
# matrix-core utilisation of the training step's GEMM / GAT kernels: one counter pass, eager launches, nothing but the kernel trace beside it
SPADOT_BENCH_NO_GRAPHS=1 timeout -k 10 400 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/mfma -- python3 $R/bench.py --steps 6 --warmup 2 --repeats 1 --no-epoch --no-cpu-baseline --leg train > /dev/null 2> $O/mfma.log
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/train -- python3 $R/bench.py --steps 40 --warmup 5 --repeats 1 --no-epoch --no-cpu-baseline --leg train > $O/train.json 2> $O/train.log
# GAT edge kernels at the cfg3 batch shape (tools/gat_bench.py, 10k nodes, bf16): HBM fetch / write bytes and L2 hits
export GAT_BENCH_REPS=6
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/gat_fetch -- python3 $R/tools/gat_bench.py > /dev/null 2> $O/gat_fetch.log
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/gat_write -- python3 $R/tools/gat_bench.py > /dev/null 2> $O/gat_write.log
timeout -k 10 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/gat_l2 -- python3 $R/tools/gat_bench.py > /dev/null 2> $O/gat_l2.log
unset GAT_BENCH_REPS
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/gat_trace -- python3 $R/tools/gat_bench.py > $O/gat_bench.txt 2> $O/gat_trace.log
cd $R
P=profiles/$TAG
python3 tools/pmc_summary.py $P/sinkhorn_cfg3_f32_pmc_summary.csv $O/pmc_fetch:FETCH_SIZE $O/pmc_write:WRITE_SIZE
python3 tools/pmc_summary.py $P/gat_cfg3_bf16_pmc_summary.csv $O/gat_fetch:FETCH_SIZE $O/gat_write:WRITE_SIZE $O/gat_l2:TCC_HIT_sum,TCC_MISS_sum --only k_gat
cp "$(ls -t $(find $O/sink -name '*kernel_stats.csv') | head -1)" $P/sinkhorn_cfg3_f32_kernel_stats.csv
cp "$(ls -t $(find $O/train -name '*kernel_stats.csv') | head -1)" $P/train_cfg3_bf16_kernel_stats.csv
cp "$(ls -t $(find $O/gat_trace -name '*kernel_stats.csv') | head -1)" $P/gat_cfg3_bf16_kernel_stats.csv
cp $O/gat_bench.txt $P/gat_cfg3_bf16_microbench.txt
tail -1 $O/sink.json > $P/sinkhorn_cfg3_f32_bench_under_rocprof.json
tail -1 $O/train.json > $P/train_cfg3_bf16_bench_under_rocprof.json
python3 tools/mfma_summary.py $O/mfma $P/train_cfg3_bf16_mfma_counters.json $P/train_cfg3_bf16_mfma_counters.csv > $O/mfma_summary.txt
python3 tools/sink_profile_meta.py $O/sink $P/sinkhorn_cfg3_f32_profile_meta.json
SPADOT_MFMA_SUMMARY=$P/train_cfg3_bf16_mfma_counters.json python3 tools/prof_summary.py $O/train 40 40 $P/train_cfg3_bf16_families.json > $P/train_cfg3_bf16_per_step_breakdown.txt
# every kernel of one step with its HIP queue, start and duration: the last step of the trace and the one three steps
# earlier (another time point, i.e. another number of inducing points)
python3 tools/prof_timeline.py $O/train 0 1 > $P/train_cfg3_bf16_step_timeline.txt
python3 tools/prof_timeline.py $O/train 0 3 > $P/train_cfg3_bf16_step_timeline_other_timepoint.txt
# GEMM evidence: the box's bare-MFMA rate, the library on the step's shapes, csrc/gemm_bf16.hip against the library
{
  echo "== tools/mfma_peak.hip (bare v_mfma loops, this box)"
  /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 tools/mfma_peak.hip -o /tmp/mfma_peak 2>/dev/null && timeout -k 10 120 /tmp/mfma_peak
  echo "== tools/gemm_shapes.py (library bf16 GEMM on the step's shapes, alone)"
  PYTHONPATH=$R timeout -k 10 200 python3 tools/gemm_shapes.py 2>/dev/null
  echo "== tools/gemm_bench.py (csrc/gemm_bf16.hip vs the library, interleaved rounds, random data)"
  PYTHONPATH=$R timeout -k 10 200 python3 tools/gemm_bench.py 2>/dev/null
  echo "== tools/gemm_bench.py wgrad (csrc/gemm_wgrad_bf16.hip vs the library's g^T x, fp32 out)"
  PYTHONPATH=$R timeout -k 10 200 python3 tools/gemm_bench.py wgrad 2>/dev/null
} > $P/gemm_probes.txt || true
# the plain line (reads the summaries just written for roofline.traffic and roofline_train)
timeout -k 10 900 python3 bench.py > $O/bench.json 2> $O/bench.log
tail -1 $O/bench.json > $P/bench_cfg3_1gpu.json
# keep gpurun_out small enough to be merged back
find $O -name '*kernel_trace.csv' -size +20M -delete
find $O -name '*counter_collection.csv' -size +20M -delete
# only gpurun_out/ travels back from the box: leave a copy of the summaries there (copy it into profiles/ afterwards)
mkdir -p $O/summaries && cp $P/* $O/summaries/
python3 tools/bench_value.py final < $P/bench_cfg3_1gpu.json
