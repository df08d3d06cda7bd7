"""The headers' ONE exactness switch (include/clover_device.h) on the CPU: dot() and threshold() start at the same end --
the reference's bits by default and under -DCLOVER_REFERENCE_BITS, the fast forms under -DCLOVER_FAST or CLV_EXACTNESS=fast -- the run-time
calls move both or one, and the older single-method macros / environment variable still override their one method."""
import os
import subprocess
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
CPP = ROOT / "tests" / "cpp"
DOT_EXACT, DOT_FAST, THR_FAST, THR_REFERENCE = 0, 1, 0, 1


def _run(tmp_path, flags=(), env_extra=None):
    obj, exe = tmp_path / "fake_clv.o", tmp_path / "exactness_switch"
    subprocess.run(["gcc", "-std=c99", "-Wall", "-c", f"-I{ROOT / 'include'}", str(CPP / "fake_clv.c"), "-o", str(obj)], check=True)
    subprocess.run(["g++", "-std=c++11", "-O1", "-Wall", "-Wextra", *flags, f"-I{ROOT / 'include'}", str(CPP / "exactness_switch.cpp"), str(obj), "-o", str(exe),
                    "-lpthread"], check=True)
    env = {k: v for k, v in os.environ.items() if k not in ("CLV_EXACTNESS", "CLV_THRESHOLD_REFERENCE")}
    env.update(env_extra or {})
    out = subprocess.run([str(exe)], env=env, check=True, capture_output=True, text=True, timeout=60).stdout
    res = {}
    for line in out.splitlines():
        what, d, t = line.split()
        res[what] = (int(d.split("=")[1]), int(t.split("=")[1]))
    return res


@pytest.mark.parametrize("flags, env, start", [
    ((), {}, (DOT_EXACT, THR_REFERENCE)),                                          # the default: the reference's bits in both
    (("-DCLOVER_REFERENCE_BITS",), {"CLV_EXACTNESS": "fast"}, (DOT_EXACT, THR_REFERENCE)),     # the macro wins over the environment
    (("-DCLOVER_FAST",), {}, (DOT_FAST, THR_FAST)),
    ((), {"CLV_EXACTNESS": "fast"}, (DOT_FAST, THR_FAST)),
    ((), {"CLV_EXACTNESS": "reference"}, (DOT_EXACT, THR_REFERENCE)),
    (("-DCLOVER_DOT_FAST",), {}, (DOT_FAST, THR_REFERENCE)),                       # the older single-method switches still move one method
    (("-DCLOVER_THRESHOLD_FAST",), {}, (DOT_EXACT, THR_FAST)),
    (("-DCLOVER_FAST", "-DCLOVER_THRESHOLD_REFERENCE"), {}, (DOT_FAST, THR_REFERENCE)),
    ((), {"CLV_THRESHOLD_REFERENCE": "0"}, (DOT_EXACT, THR_FAST)),
    (("-DCLOVER_FAST",), {"CLV_THRESHOLD_REFERENCE": "1"}, (DOT_FAST, THR_REFERENCE)),
])
def test_start_state_and_run_time_calls(tmp_path, flags, env, start):
    r = _run(tmp_path, flags, env)
    assert r["start"] == start
    assert r["set_fast"] == (DOT_FAST, THR_FAST) and r["set_reference"] == (DOT_EXACT, THR_REFERENCE)
    assert r["threshold_fast_only"] == (DOT_EXACT, THR_FAST) and r["dot_fast_only"] == (DOT_FAST, THR_REFERENCE)


def test_both_macros_together_do_not_compile(tmp_path):
    p = subprocess.run(["g++", "-std=c++11", "-fsyntax-only", "-DCLOVER_FAST", "-DCLOVER_REFERENCE_BITS", f"-I{ROOT / 'include'}",
                        str(CPP / "exactness_switch.cpp")], capture_output=True, text=True)
    assert p.returncode != 0 and "exclude each other" in p.stderr
