"""bench.py's N > 1 entry points (the split they time: CloverMatrix4.h:1700-1705).

CPU: `python bench.py --gpus N` without a launcher starts its own ranks -- the command it builds is the driver's -- and on a box with
too few GPUs it stops with a device-count message.  GPU (one MI355X): the whole N = 2 control flow runs end to end as a rehearsal
(CLOVER_BENCH_DEBUG_ONE_GPU=1: both ranks on device 0), self-launched, under an explicit torch.distributed.run, and through the
one-process clm4_sharded_* loop; plus the c5-weak preset on one GPU.  The contract keys of the JSON line are asserted each time."""
import json
import os
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
CONTRACT = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
            "data", "config", "roofline")
SMALL = ["--steps", "6", "--warmup", "2", "--rows-per-gpu", "4096", "--cols", "8192", "--no-cpu-baseline", "--no-extras"]


def run_bench(args, env_extra=None, launcher=None, timeout=600):
    env = dict(os.environ, **(env_extra or {}))
    env.pop("WORLD_SIZE", None)
    cmd = (launcher or [sys.executable]) + [str(ROOT / "bench.py"), *args]
    return subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout, cwd=str(ROOT))


def json_line(p):
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert p.returncode == 0 and len(lines) == 1, (p.returncode, p.stdout[-2000:], p.stderr[-3000:])
    out = json.loads(lines[0])
    for k in CONTRACT:
        assert k in out, k
    return out


def test_gpus2_without_enough_gpus_is_a_device_count_message():
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("this box has two GPUs")
    for extra in ([], ["--mode", "one-process"]):
        p = run_bench(["--gpus", "2", *extra], timeout=300)
        assert p.returncode != 0
        assert "GPU(s)" in p.stderr and "device-count" in p.stderr and "must be launched" not in p.stderr, p.stderr[-2000:]


def test_self_launch_builds_the_drivers_command(monkeypatch):
    """no WORLD_SIZE + --gpus 4 -> python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port P
    bench.py <the same flags>"""
    sys.path.insert(0, str(ROOT))
    import bench
    seen = {}

    class Done:
        returncode = 0

    def fake_run(cmd, env=None, **kw):
        seen["cmd"], seen["env"] = cmd, env
        return Done()
    monkeypatch.setattr(bench.subprocess, "run", fake_run)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "20", "--warmup", "5"])

    class A:
        gpus = 4
    with pytest.raises(SystemExit) as e:
        bench.self_launch(A)
    assert e.value.code == 0
    cmd = seen["cmd"]
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd and "--nproc-per-node=4" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and int(cmd[cmd.index("--master-port") + 1]) > 0
    i = cmd.index(str(ROOT / "bench.py"))
    assert cmd[i + 1:] == ["--gpus", "4", "--steps", "20", "--warmup", "5"]
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


def test_product_loops_child_failure_costs_two_objects_not_the_line(monkeypatch):
    """rank 0 runs the product's own multi-GPU loops in a child process with a time limit (bench.run_product_loops): whatever the child does
    -- exit code, no JSON, hang -- the caller gets two objects that say `failed`, and the environment the child sees carries none of the
    launcher's rank variables"""
    sys.path.insert(0, str(ROOT))
    import bench

    class A:
        steps, warmup, rows_per_gpu, cols, settle_ms, event_every, gemm_size, preset, no_c5 = 20, 5, 65536, 65536, 60.0, 2, 8192, None, False
    seen = {}

    class Bad:
        returncode, stdout, stderr = 1, "", "boom"

    def fake_run(cmd, env=None, **kw):
        seen["cmd"], seen["env"], seen["timeout"] = cmd, env, kw.get("timeout")
        return Bad()
    monkeypatch.setenv("RANK", "0")
    monkeypatch.setenv("WORLD_SIZE", "8")
    monkeypatch.setenv("LOCAL_RANK", "0")
    monkeypatch.setattr(bench.subprocess, "run", fake_run)
    out = bench.run_product_loops(A, 8)
    assert set(out) == {"one_process", "gemm_sharded"} and all("failed" in v and "exit code 1" in v["failed"] for v in out.values())
    assert "--product-loops-child" in seen["cmd"] and seen["cmd"][seen["cmd"].index("--product-loops-child") + 1] == "8"
    assert not {"RANK", "WORLD_SIZE", "LOCAL_RANK"} & set(seen["env"]) and seen["timeout"] and seen["timeout"] <= 600

    def hang(cmd, env=None, **kw):
        raise bench.subprocess.TimeoutExpired(cmd, kw.get("timeout"))
    monkeypatch.setattr(bench.subprocess, "run", hang)
    out = bench.run_product_loops(A, 8)
    assert all("TimeoutExpired" in v["failed"] for v in out.values())

    class Good:
        returncode, stderr = 0, ""
        stdout = "[noise]\n" + json.dumps({"one_process": {"ms_per_step": 1.0}, "gemm_sharded": {"value": 2.0}}) + "\n"
    monkeypatch.setattr(bench.subprocess, "run", lambda cmd, env=None, **kw: Good())
    assert bench.run_product_loops(A, 8) == {"one_process": {"ms_per_step": 1.0}, "gemm_sharded": {"value": 2.0}}


REHEARSAL = {"CLOVER_BENCH_DEBUG_ONE_GPU": "1", "CLOVER_BENCH_C5_ROWS": "16384", "CLOVER_BENCH_GEMM_SIZE": "1024"}


def check_product_loops(out, n, rehearsal, gemm_size=1024):
    """round 5: the line of a ranks run also carries the PRODUCT's own multi-GPU loops over the same N devices, driven by rank 0 behind the
    ranks' timed regions -- `one_process` (clm4_sharded_mvm_enqueue: c3 shards + its own c5) and `gemm_sharded` (configs[3] split by rows of
    A, C row panels all-gathered: SURVEY 8(e))"""
    op = out["one_process"]
    assert "failed" not in op, op
    assert op["n_gpus"] == n and op["ms_per_step"] > 0 and op["value"] > 0 and 0 < op["kernel_frac"] < 1
    assert op["gathered_result_verified"] is True and len(op["per_rank_kernel_ms"]) == n and len(op["gather_ms_behind_kernel"]) == n
    assert op["c5"]["mode"] == "one-process" and op["c5"]["gathered_result_verified"] is True and op["c5"]["n_gpus"] == n
    g = out["gemm_sharded"]
    assert "failed" not in g and "skipped" not in g, g
    assert g["n_gpus"] == n and g["rows_per_gpu"] == gemm_size // n and g["unit"] == "TOP/s" and g["value"] > 0 and g["ms_per_step"] > 0
    assert g["gathered_c_verified"] is True and len(g["per_rank_kernel_ms"]) == n and len(g["gather_ms_behind_kernel"]) == n
    assert g["kernel_only_aggregate_TOPs"] > 0 and g["c_panel_bytes"] == (gemm_size // n) * gemm_size * 4     # (sampled event pairs: no order between the two rates at toy sizes)
    assert ("NOT configs[3]" in g["workload"]) == (gemm_size != 8192)
    if rehearsal:
        assert op["degraded"] is True and g["degraded"] is True and "copies" in g["exchange"]
    return op, g


def check_c5(out, n, rows_total=16384, cols=8192):
    """the object a plain `bench.py --gpus N` adds behind the headline: BASELINE configs[4]'s matrix split N ways (here shrunk)"""
    c5 = out["c5"]
    assert c5["scaling"] == "strong" and c5["n_gpus"] == n and c5["rows_per_gpu"] == rows_total // n and c5["cols"] == cols
    assert c5["algorithmic_bytes_per_step"] == rows_total * cols // 2 + 4 * (rows_total // 64) * (cols // 64) + cols * 9 // 16 + rows_total * 9 // 16
    assert c5["ms_per_step"] > 0 and c5["value"] > 0 and 0 < c5["frac"] < 1 and 0 < c5["kernel_frac"] < 1
    assert c5["gather_bytes_per_rank"] == (rows_total // n) * 9 // 16
    assert ("configs[4]" in c5["workload"]) and (("NOT configs[4]" in c5["workload"]) == ((rows_total, cols) != (1 << 20, 65536)))
    if n > 1:
        assert c5["gathered_result_verified"] is True and len(c5["per_rank_kernel_ms"]) == n
    return c5


def check_two_way(out, mode):
    assert out["n_gpus"] == 2 and out["steps"] == 6 and out["warmup"] == 2 and out["scaling"] == "weak"
    cfg = out["config"]
    assert cfg["gathered_result_verified"] is True
    assert cfg["mode"] == mode and "DEBUG" in cfg
    assert "rccl_ranks" in cfg and "backend" in cfg and len(cfg["per_rank_kernel_ms"]) == 2
    assert out["roofline"]["bound"] == "hbm" and out["roofline"]["kernel_avg_ms"] > 0 and 0 < out["roofline"]["frac"] < 1
    assert out["value"] > 0 and out["ms_per_step"] > 0
    # a rehearsal is never a measurement: the flag a reader of a scaling curve would look at says so, at top level and in c5
    assert out["degraded"] is True and out["value_kernel_only"] > 0
    assert check_c5(out, 2)["degraded"] is True
    assert cfg["settle_launches"] >= 8
    if mode == "ranks":
        check_product_loops(out, 2, rehearsal=True)


@pytest.mark.gpu
def test_two_rank_rehearsal_self_launched():
    out = json_line(run_bench(["--gpus", "2", *SMALL], REHEARSAL))
    check_two_way(out, "ranks")
    assert out["config"]["backend"] == "gloo" and "self-launch" in out["config"]["launcher"]


@pytest.mark.gpu
def test_two_rank_rehearsal_line_carries_everything_the_8_gpu_record_needs():
    """ONE `bench.py --gpus 2` line (rehearsal: both ranks on device 0) with c5, one_process, gemm_sharded AND cpu_baseline"""
    args = [a for a in SMALL if a != "--no-cpu-baseline"] + ["--cpu-sample-rows", "1024"]
    out = json_line(run_bench(["--gpus", "2", *args], REHEARSAL))
    check_two_way(out, "ranks")
    cb = out["cpu_baseline"]
    assert cb["kind"] == "port" and cb["value"] > 0 and cb["gpu_result_matches_cpu"] is True and cb["cores"] >= 1
    assert "gemm" not in out and "extras" not in out                 # the one-GPU side measurements stay with N = 1


@pytest.mark.gpu
def test_two_rank_rehearsal_under_the_drivers_launcher():
    launcher = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                "--master-port", "29731"]
    out = json_line(run_bench(["--gpus", "2", *SMALL], REHEARSAL, launcher=launcher))
    check_two_way(out, "ranks")
    assert "external" in out["config"]["launcher"]


@pytest.mark.gpu
def test_ranks_path_through_rccl_with_one_rank():
    """CLOVER_BENCH_FORCE_DIST=1 under the launcher with ONE rank: the N > 1 code path as an 8-GPU run takes it -- gloo control group,
    RCCL data group (torch backend "nccl"), per-step all_gather_into_tensor of the packed result on the device, overlap, verification --
    executes on the one-GPU box"""
    launcher = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
                "--master-port", "29733"]
    out = json_line(run_bench(["--gpus", "1", *SMALL], {"CLOVER_BENCH_FORCE_DIST": "1", "CLOVER_BENCH_C5_ROWS": "16384", "CLOVER_BENCH_GEMM_SIZE": "1024"}, launcher=launcher))
    cfg = out["config"]
    assert out["n_gpus"] == 1 and cfg["backend"] == "nccl" and cfg["rccl_ranks"] == 1 and cfg["gathered_result_verified"] is True
    assert cfg["mode"] == "ranks" and "nccl_fallback_reason" not in cfg and "DEBUG" not in cfg
    assert cfg["gather_us_blocking"] > 0 and len(cfg["per_rank_kernel_ms"]) == 1
    assert "degraded" not in out and "degraded" not in check_c5(out, 1) and out["c5"]["backend"] == "nccl"
    assert out["ms_per_step_cold"] > 0 and out["roofline"]["kernel_avg_ms_cold"] > 0
    # the product's loops behind it, one shard, no communicator needed
    op, g = check_product_loops(out, 1, rehearsal=False)
    assert "degraded" not in op and "degraded" not in g and g["exchange"].startswith("none")


@pytest.mark.gpu
def test_product_loops_in_the_ranks_line_through_rccl_with_a_communicator_of_one_rank():
    """CLV_SHARDED_RCCL_SELFTEST=1: rank 0's one-process loops take the RCCL branches -- the in-place ncclAllGather pair of the packed mvm
    result and the ncclAllGather of the fp32 C row panel (clm4_sharded_gemm_enqueue) -- with a communicator of one rank, on the one-GPU box"""
    out = json_line(run_bench(["--gpus", "1", *SMALL], {"CLV_SHARDED_RCCL_SELFTEST": "1", "CLOVER_BENCH_C5_ROWS": "16384", "CLOVER_BENCH_GEMM_SIZE": "1024"}))
    op, g = check_product_loops(out, 1, rehearsal=False)
    assert op["rccl_ranks"] == 1 and op["backend"].startswith("rccl") and g["rccl_ranks"] == 1 and "ncclAllGather" in g["exchange"]
    assert "degraded" not in op and "degraded" not in g


@pytest.mark.gpu
def test_rccl_failure_falls_back_to_gloo_and_says_degraded():
    """the branch an 8-GPU run takes when the RCCL group cannot be built, forced: the line still appears, but `degraded` is set at top level
    and in c5, with the kernel-only aggregate beside the (host-exchange polluted) step time"""
    launcher = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
                "--master-port", "29735"]
    out = json_line(run_bench(["--gpus", "1", *SMALL], {"CLOVER_BENCH_FORCE_DIST": "1", "CLOVER_BENCH_FORCE_GLOO_FALLBACK": "1",
                                                        "CLOVER_BENCH_C5_ROWS": "16384", "CLOVER_BENCH_GEMM_SIZE": "1024"}, launcher=launcher))
    cfg = out["config"]
    assert cfg["backend"] == "gloo" and "CLOVER_BENCH_FORCE_GLOO_FALLBACK" in cfg["nccl_fallback_reason"] and cfg["gathered_result_verified"] is True
    assert out["degraded"] is True and "RCCL unavailable" in out["degraded_why"] and out["value_kernel_only"] >= out["value"] * 0.5
    c5 = check_c5(out, 1)
    assert c5["degraded"] is True and c5["backend"] == "gloo" and c5["kernel_only_aggregate_GBs"] > 0


@pytest.mark.gpu
def test_one_process_rehearsal_two_shards_on_one_gpu():
    out = json_line(run_bench(["--gpus", "2", "--mode", "one-process", *SMALL], REHEARSAL))
    check_two_way(out, "one-process")


@pytest.mark.gpu
def test_one_process_loop_through_rccl_with_a_communicator_of_one_rank():
    """the enqueue loop's RCCL branch (grouped in-place ncclAllGather pair on the exchange stream) on hardware with one GPU"""
    out = json_line(run_bench(["--gpus", "1", "--mode", "one-process", *SMALL], {"CLV_SHARDED_RCCL_SELFTEST": "1", "CLOVER_BENCH_C5_ROWS": "16384"}))
    assert out["n_gpus"] == 1 and out["config"]["rccl_ranks"] == 1 and out["config"]["gathered_result_verified"] is True
    assert out["config"]["backend"].startswith("rccl")
    c5 = check_c5(out, 1)
    assert c5["mode"] == "one-process" and c5["rccl_ranks"] == 1 and c5["gathered_result_verified"] is True and "degraded" not in out


@pytest.mark.gpu
def test_gpus1_preset_c5_weak():
    out = json_line(run_bench(["--gpus", "1", "--preset", "c5-weak", "--steps", "10", "--warmup", "3", "--no-cpu-baseline", "--no-extras"]))
    assert out["n_gpus"] == 1 and out["config"]["rows_per_gpu"] == 131072 and out["config"]["cols"] == 65536
    assert out["roofline"]["algorithmic_bytes_per_launch"] == 131072 * 65536 // 2 + 4 * 2048 * 1024 + 36864 + 73728
    assert out["roofline"]["frac"] > 0.5 and "c5" not in out          # a preset times one configuration


@pytest.mark.gpu
def test_default_line_carries_baseline_config_4():
    """plain `bench.py --gpus 1 --steps K --warmup W` (what the driver runs): c3 headline with cold + settled figures, and the c5 object =
    the whole 2^20 x 2^16 matrix (32 GiB) on this one GPU"""
    out = json_line(run_bench(["--gpus", "1", "--steps", "6", "--warmup", "2", "--no-cpu-baseline", "--no-extras"]))
    assert out["config"]["rows_per_gpu"] == 65536 and out["config"]["settle_launches"] >= 8 and "degraded" not in out
    assert out["ms_per_step_cold"] >= out["ms_per_step"] * 0.9 and out["value_cold"] > 0
    c5 = check_c5(out, 1, 1 << 20, 65536)
    assert c5["algorithmic_bytes_per_step"] == 34427473920 and c5["kernel_frac"] > 0.6
