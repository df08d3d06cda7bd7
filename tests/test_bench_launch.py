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


REHEARSAL = {"CLOVER_BENCH_DEBUG_ONE_GPU": "1"}


def check_two_way(out, mode):
    assert out["n_gpus"] == 2 and out["steps"] == 6 and out["warmup"] == 2 and out["scaling"] == "weak"
    cfg = out["config"]
    assert cfg["gathered_result_verified"] is True
    assert cfg["mode"] == mode and "DEBUG" in cfg
    assert "rccl_ranks" in cfg and "backend" in cfg and len(cfg["per_rank_kernel_ms"]) == 2
    assert out["roofline"]["bound"] == "hbm" and out["roofline"]["kernel_avg_ms"] > 0 and 0 < out["roofline"]["frac"] < 1
    assert out["value"] > 0 and out["ms_per_step"] > 0


@pytest.mark.gpu
def test_two_rank_rehearsal_self_launched():
    out = json_line(run_bench(["--gpus", "2", *SMALL], REHEARSAL))
    check_two_way(out, "ranks")
    assert out["config"]["backend"] == "gloo" and "self-launch" in out["config"]["launcher"]


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
    out = json_line(run_bench(["--gpus", "1", *SMALL], {"CLOVER_BENCH_FORCE_DIST": "1"}, launcher=launcher))
    cfg = out["config"]
    assert out["n_gpus"] == 1 and cfg["backend"] == "nccl" and cfg["rccl_ranks"] == 1 and cfg["gathered_result_verified"] is True
    assert cfg["mode"] == "ranks" and "nccl_fallback_reason" not in cfg and "DEBUG" not in cfg
    assert cfg["gather_us_blocking"] > 0 and len(cfg["per_rank_kernel_ms"]) == 1


@pytest.mark.gpu
def test_one_process_rehearsal_two_shards_on_one_gpu():
    out = json_line(run_bench(["--gpus", "2", "--mode", "one-process", *SMALL], REHEARSAL))
    check_two_way(out, "one-process")


@pytest.mark.gpu
def test_one_process_loop_through_rccl_with_a_communicator_of_one_rank():
    """the enqueue loop's RCCL branch (grouped in-place ncclAllGather pair on the exchange stream) on hardware with one GPU"""
    out = json_line(run_bench(["--gpus", "1", "--mode", "one-process", *SMALL], {"CLV_SHARDED_RCCL_SELFTEST": "1"}))
    assert out["n_gpus"] == 1 and out["config"]["rccl_ranks"] == 1 and out["config"]["gathered_result_verified"] is True
    assert out["config"]["backend"].startswith("rccl")


@pytest.mark.gpu
def test_gpus1_preset_c5_weak():
    out = json_line(run_bench(["--gpus", "1", "--preset", "c5-weak", "--steps", "10", "--warmup", "3", "--no-cpu-baseline", "--no-extras"]))
    assert out["n_gpus"] == 1 and out["config"]["rows_per_gpu"] == 131072 and out["config"]["cols"] == 65536
    assert out["roofline"]["algorithmic_bytes_per_launch"] == 131072 * 65536 // 2 + 4 * 2048 * 1024 + 36864 + 73728
    assert out["roofline"]["frac"] > 0.5
