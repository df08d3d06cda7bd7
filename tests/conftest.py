import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle.binding import Oracle
    return Oracle()


@pytest.fixture(scope="session")
def fast_oracle():
    from oracle.binding import FastOracle
    return FastOracle()


@pytest.fixture(scope="session")
def hip():
    """The HIP library on a real device.  No fallback: a missing .so or GPU is an error, not a skip."""
    from clover_amd.lib_binding import CloverHip
    return CloverHip()


def bits(a):
    return np.asarray(a, dtype=np.float32).view(np.uint32)


def kat2_inputs():
    i = np.arange(256)
    x = (((37 * i) % 101) - 50).astype(np.float32) * np.float32(0.125)
    y = (((53 * i) % 89) - 44).astype(np.float32) * np.float32(0.0625)
    return x, y


def kat3_inputs():
    M, N = 128, 256
    r, c = np.meshgrid(np.arange(M), np.arange(N), indexing="ij")
    A = (((31 * r + 17 * c) % 23) - 11).astype(np.float32)
    x = (((13 * np.arange(N)) % 19) - 9).astype(np.float32)
    return A, x


def random_packed(rng, n):
    """n elements of random nibbles in [-7,7] (packed) + positive scales."""
    q = rng.integers(-7, 8, size=n).astype(np.int8)
    b = ((q[0::2].astype(np.uint8) & 0xF) << 4) | (q[1::2].astype(np.uint8) & 0xF)
    s = rng.uniform(0.5, 2.0, size=n // 64).astype(np.float32)
    return b.astype(np.uint8), s
