"""The fp32 classes' own arithmetic on the host (include/clover_fp32.h; CloverVector32.h:160-684, CloverMatrix32.h:90-215): the baseline the
reference compares every 4-bit result with.  CPU only (tests/cpp/fake_clv.c stands in for the C ABI).  dot is checked bit for bit against an
AVX2 + FMA evaluation in the reference's order; the survivors of a tie-heavy threshold against the oracle's heap walk (pinned on the real
std::make_heap, tests/cpp/threshold_stdheap.cpp) run over a 4-bit vector that holds the same magnitudes."""
import subprocess
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent
CPP = ROOT / "tests" / "cpp"


@pytest.fixture(scope="module")
def exe(tmp_path_factory):
    d = tmp_path_factory.mktemp("fp32")
    obj, out = d / "fake_clv.o", d / "fp32_baseline"
    subprocess.run(["gcc", "-std=c99", "-Wall", "-c", f"-I{ROOT / 'include'}", str(CPP / "fake_clv.c"), "-o", str(obj)], check=True)
    subprocess.run(["g++", "-std=c++11", "-O2", "-Wall", "-Wextra", "-mavx2", "-mfma", "-ffp-contract=off", "-fopenmp", f"-I{ROOT / 'include'}",
                    str(CPP / "fp32_baseline.cpp"), str(obj), "-o", str(out), "-lpthread"], check=True)
    return out


def test_fp32_methods(exe):
    p = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    assert p.returncode == 0 and p.stdout.strip().endswith("ok"), p.stdout[-2000:]


@pytest.mark.parametrize("n, k", [(128, 64), (1000, 250), (4096, 1), (5000, 4999), (777, 300), (8192, 2048)])
def test_fp32_threshold_survivors_match_the_heap_walk(exe, oracle, n, k):
    p = subprocess.run([str(exe), str(n), str(k)], capture_output=True, text=True, timeout=120)
    assert p.returncode == 0, p.stdout[-2000:]
    lines = {ln.split()[0]: np.array(ln.split()[1:], dtype=np.int32) for ln in p.stdout.splitlines() if ln.startswith(("values", "survivors"))}
    vals, surv = lines["values"], lines["survivors"]
    assert vals.size == n and np.abs(vals).max() <= 7
    # the same magnitudes as a CloverVector4: nibbles = the integers, every scale 7 -> get(i) = (7 / 7) * q
    npad = (n + 127) // 128 * 128
    q = np.zeros(npad, np.int32)
    q[:n] = vals
    packed = (((q[0::2] & 0xF) << 4) | (q[1::2] & 0xF)).astype(np.uint8)
    s = np.full(npad // 64, 7.0, np.float32)
    ref = oracle.v4_threshold(packed, s, n, k)
    hi, lo = (ref.astype(np.int8) >> 4).astype(np.int32), ((ref << 4).astype(np.int8) >> 4).astype(np.int32)
    ref_vals = np.stack([hi, lo], 1).reshape(-1)[:n]
    assert np.array_equal(surv, ref_vals)
    assert int((surv != 0).sum()) <= k
