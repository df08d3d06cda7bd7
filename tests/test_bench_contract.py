"""bench.py keeps the driver's contract: one JSON line with the agreed keys (small shapes here; the real run is the driver's)."""
import json
import os
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
CONTRACT = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
            "dtype", "data", "config", "roofline")


def _run(args):
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)
    out = subprocess.run([sys.executable, str(ROOT / "bench.py"), *args], cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, out.stdout
    return json.loads(lines[0])


def test_help_runs_without_a_gpu():
    out = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--help"], cwd=ROOT, capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and "--gpus" in out.stdout and "--steps" in out.stdout and "--warmup" in out.stdout


@pytest.mark.gpu
def test_mvm_line():
    d = _run(["--gpus", "1", "--steps", "3", "--warmup", "1", "--rows-per-gpu", "8192", "--cols", "8192", "--cpu-sample-rows", "1024", "--no-c5"])
    for k in CONTRACT + ("cpu_baseline",):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1 and d["unit"] == "GB/s" and d["higher_is_better"] is True
    assert d["value"] > 0 and d["ms_per_step"] > 0 and d["scaling"] == "weak" and d["dtype"] == "int4" and d["data"] == "synthetic"
    assert "workload" in d["config"] and "model" not in d["config"]
    rf = d["roofline"]
    assert rf["bound"] == "hbm" and rf["unit"] == "GB/s" and rf["peak"] == 8000.0 and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["unit"] == "GB/s" and cb["gpu_result_matches_cpu"] is True
    assert cb["ms_min"] <= cb["ms"] <= cb["ms_max"] and cb["runnable_cpus"] >= 1 and cb["threads_used"] == cb["cores"]
    assert cb["within_quota"]["GB/s"] > 0 and str(cb["within_quota"]["threads"]) in cb["by_threads"]
    # round 5: under a cgroup cpu quota `value` is the SUSTAINED team (within the quota); the best burst stays beside it
    assert cb["burst_best"]["value"] >= cb["value"] * 0.999 and cb["cores"] == cb["threads_used"]
    assert cb["value"] == (cb["within_quota"]["GB/s"] if cb["quota_cpus"] else cb["burst_best"]["value"])
    assert d["ms_per_step_cold"] > 0 and d["config"]["settle_launches"] >= 8
    # the second roofline-bearing object: BASELINE configs[3], normalised to the pipe the kernel runs on
    g = d["gemm"]
    assert g["roofline"]["bound"] == "mfma" and g["roofline"]["peak"] == 10000.0 and abs(g["roofline"]["frac"] - g["value"] / 10000.0) < 1e-3
    assert g["prepared_operands"]["ms"] > 0 and g["int32_unscaled"]["ms"] > 0
    # round 3: the ceiling of the definition measured on this box (probe library, child processes), the int8-MFMA kernel, counters
    c = g["ceiling"]
    assert 0 < c["mfma_only_ms"] < c["mfma_plus_fold_only_ms"] < c["full_loop_ms"] and 0 < c["frac_if_only_arithmetic"] < 1
    assert g["int8_mfma_kernel"]["ms"] > g["ms"] * 0.5
    assert g["roofline"]["traffic"] > g["roofline"]["algorithmic_bytes_per_call"] and 0 < g["roofline"]["mfma_busy_pct"] < 100
    # BASELINE configs[1] on the host: quantize and dot, one core and all cores, cache-sized and DRAM-sized
    v = cb["vector_ops"]
    for size in ("n2^24", "n2^30"):
        for op in ("quantize_sequential", "quantize_all_cores", "dot_sequential", "dot_all_cores"):
            assert v[size][op]["value"] > 0 and v[size][op]["threads"] >= 1, (size, op)
    assert d["extras"]["footnote_cache_resident_n2^24"]["dot_exact"]["host_one_core_ms_same_order"] > 0
    assert "kernel_avg_of" in rf
    # HBM-resident vector workloads, each with its own achieved / peak / frac
    h = d["extras"]["hbm_resident_n2^30"]
    for k in ("quantize", "quantize_stochastic", "dot_fast", "scale_and_add", "restore"):
        assert h[k]["peak"] == 8000.0 and 0 < h[k]["frac"] < 1 and abs(h[k]["frac"] - h[k]["achieved"] / 8000.0) < 1e-3


@pytest.mark.gpu
def test_gemm_line():
    d = _run(["--workload", "gemm", "--gemm-size", "1024", "--steps", "3", "--warmup", "1"])
    for k in CONTRACT:
        assert k in d, k
    assert d["unit"] == "TOP/s" and d["roofline"]["bound"] == "mfma" and d["value"] > 0
