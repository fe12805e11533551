"""bench.py's ONE stdout line: size, required keys, and the `--gpus N` launcher.

The r03 line carried a 26 KB per-kernel table and outgrew the window the driver keeps of stdout, so the round's
headline was unparsed.  These tests build the line from a canned full record (the r03 run's own record, committed
under profiles/r03/) and rehearse the launcher on CPU (gloo), with no GPU call anywhere."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def canned_record():
    d = json.load(open(os.path.join(ROOT, "profiles", "r03", "bench_cfg3_1gpu.json")))
    tbl = d["roofline_train"].pop("mfma_counters")
    d["roofline_train"]["mfma_busy_top"] = bench.mfma_top(tbl)
    return d, tbl


def test_line_is_short_and_complete():
    full, _ = canned_record()
    o, line = bench.compact_line(full, "gpurun_out/bench_detail_1gpu.json")
    assert len(line) < bench.LINE_LIMIT and "\n" not in line
    back = json.loads(line)
    for k in bench.REQUIRED_KEYS:
        assert k in back, k
    assert isinstance(back["config"]["workload"], str) and "model" not in back["config"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "alg_bytes_per_launch"):
        assert k in back["roofline"], k
    assert abs(back["roofline"]["frac"] - back["roofline"]["achieved"] / back["roofline"]["peak"]) < 1e-5
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in back["cpu_baseline"], k
    assert back["cpu_baseline"]["kind"] in ("port", "reference")
    assert abs(back["value"] - full["value"]) <= 1e-5 * full["value"]
    assert abs(back["ms_per_step"] - full["ms_per_step"]) <= 1e-5 * full["ms_per_step"]
    assert back["detail_file"].endswith("bench_detail_1gpu.json")


def test_mfma_summary_keeps_at_most_four_kernels():
    _, tbl = canned_record()
    top = bench.mfma_top(tbl)
    assert 1 <= len(top) <= 4 and all(0.0 <= v <= 1.0 for v in top.values())
    assert len(json.dumps(top)) < 400


def test_an_overlong_line_fails_loudly():
    full, tbl = canned_record()
    full["config"]["workload"] = "x" * 7000
    with pytest.raises(RuntimeError, match="limit"):
        bench.compact_line(full)


def test_multi_rank_line_fits_too():
    full, _ = canned_record()
    full["n_gpus"] = 8
    full["rank_devices"] = [f"{i}:AMD Instinct MI355X" for i in range(8)]
    full["config"]["parallelism"] = "dp8: " + "y" * 600
    full.pop("cpu_baseline")
    _, line = bench.compact_line(full, "gpurun_out/bench_detail_8gpu.json")
    assert len(line) < bench.LINE_LIMIT


def _run(args, env_extra, timeout=240):
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    env.update(env_extra)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, stdout=subprocess.PIPE,
                          stderr=subprocess.PIPE, text=True, timeout=timeout)


def test_gpus_n_starts_n_ranks_and_prints_one_line():
    """`python bench.py --gpus 2` without WORLD_SIZE: the launcher branch starts 2 ranks (gloo here), rank 0 prints ONE line
    that says n_gpus = 2 and the all-reduce of ones saw 2 ranks."""
    r = _run(["--gpus", "2", "--rendezvous-only"], {"SPADOT_BENCH_BACKEND": "gloo"})
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["rccl_ranks"] == 2 and rec["backend"] == "gloo"


def test_world_size_that_disagrees_with_gpus_is_refused():
    """A rank environment whose WORLD_SIZE is not --gpus must not print a line (it would carry the wrong n_gpus)."""
    r = _run(["--gpus", "3", "--no-cpu-baseline"], {"WORLD_SIZE": "1"}, timeout=120)
    assert r.returncode == 4 and r.stdout.strip() == "", (r.returncode, r.stdout, r.stderr[-500:])
