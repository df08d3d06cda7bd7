"""tests/cpp/sharded_mvm.cpp: the one-process multi-GPU path (clm4_sharded_*) through the C ABI -- with the GPUs the box has
(1 here, 8 on a full node: RCCL all-gather) and, on device 0, with the 3/5/7/8-way partitions incl. odd 64-row shards."""
import subprocess

import numpy as np
import pytest

from clover_amd.build import build_hip_library, repo_root

ROOT = repo_root()


def _build(tmp_path):
    lib = build_hip_library()
    exe = tmp_path / "sharded_mvm"
    subprocess.run(["g++", "-std=c++11", "-O1", "-Wall", "-Wextra", f"-I{ROOT / 'include'}", str(ROOT / "tests" / "cpp" / "sharded_mvm.cpp"), "-o", str(exe),
                    f"-L{lib.parent}", "-lclover_hip", f"-Wl,-rpath,{lib.parent}", "-L/opt/rocm/lib", "-Wl,-rpath,/opt/rocm/lib"], check=True)
    return exe


def test_sharded_cpp_client_builds_and_reports_no_device_on_cpu(tmp_path):
    p = subprocess.run([str(_build(tmp_path))], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0 and ("no_device" in p.stdout or "sharded all ok" in p.stdout), (p.returncode, p.stdout, p.stderr)


@pytest.mark.gpu
def test_sharded_cpp_all_layouts_equal_unsharded(tmp_path):
    p = subprocess.run([str(_build(tmp_path))], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0 and "sharded all ok" in p.stdout, (p.returncode, p.stdout, p.stderr)
    lines = {ln.split()[0]: ln for ln in p.stdout.splitlines() if " ok" in ln and "parts=" in ln}
    assert set(lines) == {"node", "loop8", "loop3", "loop5", "loop7"}
    assert "odd_shards=0" not in lines["loop3"] and "equal=0" in lines["loop3"] and "equal=1" in lines["loop8"]
    print(p.stdout)


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["1", "ragged"])
def test_sharded_cpp_through_rccl_with_a_communicator_of_one_rank(tmp_path, mode):
    """CLV_SHARDED_RCCL_SELFTEST: the node layout (one GPU here) takes the RCCL path -- librccl dlopen'ed, ncclCommInitAll, the
    grouped broadcast of x, the in-place ncclAllGather pair ("1") or the per-owner broadcasts ("ragged") -- with a communicator of
    one rank, and must still equal the unsharded call.  What an 8-GPU node adds is more ranks, not other calls."""
    import os
    env = dict(os.environ, CLV_SHARDED_RCCL_SELFTEST=mode)
    p = subprocess.run([str(_build(tmp_path))], capture_output=True, text=True, timeout=600, env=env)
    assert p.returncode == 0 and "sharded all ok" in p.stdout, (p.returncode, p.stdout, p.stderr)
    node = [ln for ln in p.stdout.splitlines() if ln.startswith("node ")][0]
    assert "rccl_ranks=1" in node and (f"equal={1 if mode == '1' else 0}" in node), node


@pytest.mark.gpu
@pytest.mark.parametrize("blocks", [1, 3, 7, 13])
def test_mvm_family_accepts_odd_64_row_shards(hip, oracle, blocks):
    """a shard with an odd number of 64-row blocks (rows % 128 != 0) must go through clm4_mvm / clm4_rowdots /
    clm4_mvm_f32 / clm4_mvm_v8 and equal the same rows of the whole-matrix result."""
    rs = np.random.default_rng(blocks)
    total, cols = 64 * (blocks + 3), 640
    total += total % 128
    A = rs.integers(-10, 11, size=(total, cols)).astype(np.float32)
    qA, sA = hip.m4_quantize(A)
    x = rs.integers(-10, 11, size=cols).astype(np.float32)
    qx, sx = hip.v4_quantize(x)
    r, sr = hip.m4_mvm(qA, sA, total, cols, qx, sx)
    d = hip.m4_rowdots(qA, sA, total, cols, qx, sx)
    f = hip.m4_mvm_f32(qA, sA, total, cols, x)
    q8, s8 = hip.v8_quantize(x)
    r8, sr8 = hip.m4_mvm_v8(qA, sA, total, cols, q8, s8)
    b0 = 1                                               # shard = blocks b0 .. b0+blocks-1
    rows = 64 * blocks
    qs, ss = qA[b0 * 64 * cols // 2:(b0 + blocks) * 64 * cols // 2], sA[b0 * (cols // 64):(b0 + blocks) * (cols // 64)]
    rr, srr = hip.m4_mvm(qs, ss, rows, cols, qx, sx)
    assert np.array_equal(rr, r[b0 * 32:(b0 + blocks) * 32]) and np.array_equal(srr.view(np.uint32), sr[b0:b0 + blocks].view(np.uint32))
    ro, sro = oracle.m4_mvm(qs, ss, rows, cols, qx, sx)
    assert np.array_equal(rr, ro) and np.array_equal(srr.view(np.uint32), sro.view(np.uint32))
    assert np.array_equal(hip.m4_rowdots(qs, ss, rows, cols, qx, sx).view(np.uint32), d[b0 * 64:(b0 + blocks) * 64].view(np.uint32))
    assert np.array_equal(hip.m4_mvm_f32(qs, ss, rows, cols, x).view(np.uint32), f[b0 * 64:(b0 + blocks) * 64].view(np.uint32))
    rr8, srr8 = hip.m4_mvm_v8(qs, ss, rows, cols, q8, s8)
    assert np.array_equal(rr8, r8[b0 * 64:(b0 + blocks) * 64]) and np.array_equal(srr8.view(np.uint32), sr8[b0:b0 + blocks].view(np.uint32))
