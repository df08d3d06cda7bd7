"""The C++ drop-in surface: user code in the reference's style compiles against include/ (CPU check) and,
on the GPU box, produces the reference's known answers through libclover_hip.so."""
import json
import subprocess
from pathlib import Path

import numpy as np
import pytest

from clover_amd.build import build_hip_library, repo_root
from conftest import bits, kat3_inputs

ROOT = repo_root()
SRC = ROOT / "tests" / "cpp" / "dropin_example.cpp"
KAT = json.loads((Path(__file__).parent / "golden" / "kat_reference.json").read_text())


def compile_example(out: Path, extra=()):
    lib = build_hip_library()
    cmd = ["g++", "-std=c++11", "-O1", "-Wall", "-Wextra", "-DCLOVER_STOCHASTIC_ROUNDING_DISABLED=1", f"-I{ROOT / 'include'}",
           str(SRC), "-o", str(out), f"-L{lib.parent}", "-lclover_hip", f"-Wl,-rpath,{lib.parent}",
           "-L/opt/rocm/lib", "-Wl,-rpath,/opt/rocm/lib", *extra]
    subprocess.run(cmd, check=True)


def test_dropin_example_compiles_and_links(tmp_path):
    # host language of the reference is C++11 (CMakeLists.txt:44-76); the headers must build with it
    compile_example(tmp_path / "dropin")
    for hdr in ("CloverVector32.h", "CloverVector4.h", "CloverVector8.h", "CloverMatrix32.h", "CloverMatrix4.h", "CloverIHT.h"):
        subprocess.run(["g++", "-std=c++11", "-fsyntax-only", "-x", "c++", f"-I{ROOT / 'include'}", str(ROOT / "include" / hdr)], check=True)


@pytest.mark.gpu
@pytest.mark.parametrize("exactness", ["reference_bits", "fast"])
def test_dropin_example_matches_reference_answers(tmp_path, oracle, exactness):
    """the README-style client in both settings of the headers' ONE exactness switch (clover_device.h): the default build
    (= -DCLOVER_REFERENCE_BITS: dot() in the reference's order, threshold() the reference's heap walk -- Q_IHT through CloverIHT.h
    must follow the ORACLE's loop, whose threshold is the reference's, tie for tie) and -DCLOVER_FAST (lowest-index ties)."""
    exe = tmp_path / "dropin"
    compile_example(exe, extra=("-DCLOVER_FAST",) if exactness == "fast" else ())
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True, timeout=300).stdout
    kv = {}
    for line in out.splitlines():
        for tok in line.split():
            if "=" in tok:
                k, v = tok.split("=", 1)
                kv[k] = v
    assert kv["kat1_dot"] == KAT["KAT1"]["dot_bits"] and kv["kat1_dot_scalar"] == KAT["KAT1"]["dot_scalar_bits"]
    assert float(kv["kat1_dot_parallel"]) == 256.0 and kv["kat1_bytes"] == "77" * 8 and kv["kat1_scales"] == "1,2"
    assert kv["kat1_get"] == "2" and kv["bytes"] == "72"
    assert kv["kat2_qx"] == KAT["KAT2"]["qx_bytes_0_31"] and kv["kat2_qy"] == KAT["KAT2"]["qy_bytes_0_31"]
    assert kv["kat2_dot_scalar"] == KAT["KAT2"]["dot_scalar_bits"]
    if exactness == "fast":       # dot() is the fast order under -DCLOVER_FAST: exact block integers, other fp32 order (tolerance as test_gpu_parity)
        want = float(np.array([int(KAT["KAT2"]["dot_bits"], 16)], np.uint32).view(np.float32)[0])
        for key in ("kat2_dot", "kat2_copy_dot", "kat2_view_dot"):
            got = float(np.array([int(kv[key], 16)], np.uint32).view(np.float32)[0])
            assert abs(got - want) <= 1e-5 * abs(want), (key, got, want)
    else:
        assert kv["kat2_dot"] == KAT["KAT2"]["dot_bits"]
        assert kv["kat2_copy_dot"] == KAT["KAT2"]["dot_bits"] and kv["kat2_view_dot"] == KAT["KAT2"]["dot_bits"]
    assert kv["kat2_restore"].split(",") == KAT["KAT2"]["restore_qx_0_3_bits"]
    assert kv["kat3_r"] == KAT["KAT3"]["r_bytes_0_63"] and kv["kat3_r_parallel"] == KAT["KAT3"]["r_bytes_0_63"]
    assert kv["kat3_scales"].split(",") == KAT["KAT3"]["r_scale_bits"]
    np.testing.assert_allclose([float(v) for v in kv["kat3_get"].split(",")], KAT["KAT3"]["qA_get_0_0_3"], atol=1e-5)
    head = [ln for ln in out.splitlines() if ln.startswith("kat3_tostring_head=")][0].split("=", 1)[1]
    assert head.rstrip() == " -11.00    4.71    0.00   -4.71"                    # -11, 4.71429, 0, -4.71429 with setw(7), precision 2
    # mixed precision: CloverVector8 + CloverMatrix4::mvm(CloverVector8, CloverVector8) against the oracle
    A3, _ = kat3_inputs()
    x3 = np.array([np.float32(float(((13 * c) % 19) - 9)) * np.float32(0.37) for c in range(256)], np.float32)
    q4, s4 = oracle.m4_quantize(A3)
    qx8, sx8 = oracle.v8_quantize(x3)
    r8, sr8 = oracle.m4_mvm_v8(q4, s4, 128, 256, qx8, sx8)
    assert kv["mixed_x8"] == qx8[:64].tobytes().hex() and kv["mixed_r8"] == r8.tobytes().hex()
    assert [int(v, 16) for v in kv["mixed_scales"].split(",")] == [int(bits(sr8[0])), int(bits(sr8[1]))] and kv["bytes8"] == str(256 + 4 * 4)
    back = oracle.v8_restore(qx8, sx8)
    assert [int(v, 16) for v in kv["mixed_restore"].split(",")] == [int(bits(back[1])), int(bits(back[255]))]
    # CloverVector8::dot through the header (round 5): x8 . x8
    d8 = oracle.v8_dot(qx8, sx8, qx8, sx8)
    if exactness == "fast":
        got8 = float(np.array([int(kv["mixed_dot8"], 16)], np.uint32).view(np.float32)[0])
        assert abs(got8 - float(d8)) <= 1e-5 * abs(float(d8))
    else:
        assert int(kv["mixed_dot8"], 16) == int(bits(d8))
    assert abs(float(kv["dot8_parallel"]) - float(d8)) <= 1e-5 * abs(float(d8))
    assert abs(float(kv["getabs"]) - abs(float(back[1]))) <= 1e-7 * max(1.0, abs(float(back[1])))
    assert int(kv["get"], 16) == int(bits(np.float32(np.float32(np.float32(qx8[1]) * sx8[0]) / np.float32(127.0))))
    # the IHT-style iteration built from the "next" rows
    assert kv["iht_transpose_ok"] == "1" and 0 < int(kv["iht_nonzeros"]) <= 32
    # Q_IHT / Q_GD (CloverIHT.h) against the same loops on the oracle.  Data = the C++ setRandomInteger streams.
    def ints(n, mx, seed):
        z, out, m = seed, np.zeros(n, np.float32), (1 << 64) - 1
        for i in range(n):
            z = (z + 0x9E3779B97F4A7C15) & m
            r = z
            r = ((r ^ (r >> 30)) * 0xBF58476D1CE4E5B9) & m
            r = ((r ^ (r >> 27)) * 0x94D049BB133111EB) & m
            r ^= r >> 31
            out[i] = float(int(r % (2 * mx + 1)) - mx)
        return out
    M, N, K = 256, 512, 32
    Phi = oracle.m4_quantize(ints(M * N, 10, 7).reshape(M, N))
    PhiT = oracle.m4_transpose(*Phi, M, N)
    y = oracle.v4_quantize(ints(M, 10, 9))

    def threshold_lowest_index(q, s, k):          # the GPU's tie rule: > tau, then first ties
        mags = np.abs(oracle.v4_restore(q, s))
        tau = np.sort(mags)[::-1][k - 1]
        keep = mags > tau
        ties = np.flatnonzero(mags == tau)[: k - int(keep.sum())]
        keep[ties] = True
        hi = (q.astype(np.int8) >> 4).astype(np.int32)
        lo = ((q << 4).astype(np.int8) >> 4).astype(np.int32)
        nib = np.stack([hi, lo], 1).reshape(-1) * keep
        return (((nib[0::2] & 0xF) << 4) | (nib[1::2] & 0xF)).astype(np.uint8)

    def loop(iters, thr):
        x = (np.zeros(N // 2, np.uint8), np.ones(N // 64, np.float32))       # x.clear()
        for _ in range(iters):
            t1 = oracle.m4_mvm(*Phi, M, N, *x)
            t2 = oracle.v4_scale_and_add(*y, *t1, -1.0)
            t3 = oracle.m4_mvm(*PhiT, N, M, *t2)
            x = oracle.v4_scale_and_add(*x, *t3, 0.001)
            if thr:
                # default build: the reference's survivors (orc_v4_threshold = its min-heap walk, CloverVector4.h:1913-2060)
                x = ((threshold_lowest_index(x[0], x[1], K) if exactness == "fast" else oracle.v4_threshold(x[0], x[1], N, K)), x[1])
        return x
    xi = loop(3, True)
    assert kv["qiht_x"] == xi[0].tobytes().hex()
    assert kv["qiht_scales"].split(",") == [hex(bits(xi[1][0])), hex(bits(xi[1][7]))]
    assert kv["qgd_x"] == loop(2, False)[0].tobytes().hex()
    # the mixed-precision loops (CloverMatrix4 + CloverVector8), same operands
    y8 = oracle.v8_quantize(ints(M, 10, 9))

    def threshold8_lowest_index(q, s, k):
        mags = np.abs((q.astype(np.float32) * np.repeat(s, 64)) / np.float32(127.0))
        tau = np.sort(mags)[::-1][k - 1]
        keep = mags > tau
        ties = np.flatnonzero(mags == tau)[: k - int(keep.sum())]
        keep[ties] = True
        return (q * keep).astype(np.int8)

    def loop8(iters, thr):
        x = (np.zeros(N, np.int8), np.ones(N // 64, np.float32))
        for _ in range(iters):
            t1 = oracle.m4_mvm_v8(*Phi, M, N, *x)
            t2 = oracle.v8_scale_and_add(*y8, *t1, -1.0)
            t3 = oracle.m4_mvm_v8(*PhiT, N, M, *t2)
            x = oracle.v8_scale_and_add(*x, *t3, 0.001)
            if thr:
                x = ((threshold8_lowest_index(x[0], x[1], K) if exactness == "fast" else oracle.v8_threshold(x[0], x[1], N, K)), x[1])
        return x
    x8 = loop8(3, True)
    assert kv["qiht8_x"] == x8[0].tobytes().hex()
    assert [int(v, 16) for v in kv["qiht8_scales"].split(",")] == [int(bits(x8[1][0])), int(bits(x8[1][7]))]
    assert kv["qgd8_x"] == loop8(2, False)[0].tobytes().hex()
    # GEMM spot values vs the oracle's definition
    A, _ = kat3_inputs()
    qA, sA = oracle.m4_quantize(A)
    C = oracle.m4_gemm(qA, sA, 128, 256, qA, sA, 128)
    assert kv["gemm_c00"] == hex(bits(C[0, 0])) or int(kv["gemm_c00"], 16) == int(bits(C[0, 0]))
    assert int(kv["gemm_c_1_77"], 16) == int(bits(C[1, 77]))


def test_error_convention_and_layout_without_gpu(tmp_path):
    """message + exit(1) on shape errors, like the reference; host layout contract of the containers"""
    lib = build_hip_library()
    exe = tmp_path / "errs"
    subprocess.run(["g++", "-std=c++11", "-O1", "-Wall", f"-I{ROOT / 'include'}", str(ROOT / "tests" / "cpp" / "error_behaviour.cpp"),
                    "-o", str(exe), f"-L{lib.parent}", "-lclover_hip", f"-Wl,-rpath,{lib.parent}", "-L/opt/rocm/lib",
                    "-Wl,-rpath,/opt/rocm/lib"], check=True)
    for case, msg in (("mvm", "MVM can not be performed. Exiting ..."), ("mvm8", "MVM can not be performed. Exiting ..."), ("quantize", "Matrices do not have the same size. Exiting ..."),
                      ("transpose", "Matrix can not be transposed. Exiting ...")):
        p = subprocess.run([str(exe), case], capture_output=True, text=True)
        assert p.returncode == 1 and msg in p.stdout and "not reached" not in p.stdout
    p = subprocess.run([str(exe), "layout"], capture_output=True, text=True)
    assert p.returncode == 0 and "layout ok" in p.stdout, (p.returncode, p.stdout)


def _compile_gemm_cache(out: Path, extra=()):
    lib = build_hip_library()
    subprocess.run(["g++", "-std=c++11", "-O1", "-Wall", "-Wextra", "-DCLOVER_STOCHASTIC_ROUNDING_DISABLED=1", f"-I{ROOT / 'include'}",
                    str(ROOT / "tests" / "cpp" / "gemm_cache.cpp"), "-o", str(out), f"-L{lib.parent}", "-lclover_hip", f"-Wl,-rpath,{lib.parent}",
                    "-L/opt/rocm/lib", "-Wl,-rpath,/opt/rocm/lib", *extra], check=True)


def test_gemm_cache_client_builds_and_reports_no_device_on_cpu(tmp_path):
    exe = tmp_path / "gemm_cache"
    _compile_gemm_cache(exe)
    p = subprocess.run([str(exe)], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0 and ("no_device" in p.stdout or "gemm_cache ok" in p.stdout), (p.returncode, p.stdout, p.stderr)


@pytest.mark.gpu
@pytest.mark.parametrize("tracking", [True, False])
def test_gemm_operand_cache_follows_the_matrix(tmp_path, tracking):
    """CloverMatrix4::cacheGemmOperand: the FP6 image kept between gemm() calls is rebuilt after quantize() and after a write
    through a kept getData() pointer (page tracking) -- and, without page tracking, after every getData() -- and never changes a
    result."""
    exe = tmp_path / "gemm_cache"
    _compile_gemm_cache(exe, () if tracking else ("-DCLOVER_HIP_NO_PAGE_TRACKING",))
    p = subprocess.run([str(exe)], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0 and "gemm_cache ok" in p.stdout, (p.returncode, p.stdout, p.stderr)


@pytest.mark.gpu
@pytest.mark.parametrize("explicit", [False, True])
def test_threshold_min_heap_hands_back_the_reference_heap(tmp_path, explicit):
    """CloverVector4 / CloverVector8 ::threshold_min_heap(idx_t *, k) (CloverVector4.h:1929-1970): the caller's heap memory receives the
    reference's heap entry for entry (value, bits, idx) -- compared with the reference's method restated over std::make_heap on the host --
    and the vector keeps exactly the heap's elements; both container builds"""
    lib = build_hip_library()
    exe = tmp_path / "threshold_min_heap"
    subprocess.run(["g++", "-std=c++11", "-O1", "-Wall", "-Wextra", "-DCLOVER_STOCHASTIC_ROUNDING_DISABLED=1", *(["-DCLOVER_HIP_EXPLICIT_SYNC"] if explicit else []),
                    f"-I{ROOT / 'include'}", str(ROOT / "tests" / "cpp" / "threshold_min_heap.cpp"), "-o", str(exe), f"-L{lib.parent}", "-lclover_hip",
                    f"-Wl,-rpath,{lib.parent}", "-L/opt/rocm/lib", "-Wl,-rpath,/opt/rocm/lib", "-lpthread"], check=True)
    p = subprocess.run([str(exe)], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0 and "threshold_min_heap OK" in p.stdout, (p.returncode, p.stdout, p.stderr)


# ---- explicit residency (-DCLOVER_HIP_EXPLICIT_SYNC, include/clover_device.h): no signal handler, no mprotect -----------------------
def _build_explicit(tmp_path, explicit: bool):
    lib = build_hip_library()
    exe = tmp_path / ("explicit_sync" if explicit else "explicit_sync_tracked")
    subprocess.run(["g++", "-std=c++11", "-O2", "-Wall", "-Wextra", "-DCLOVER_STOCHASTIC_ROUNDING_DISABLED=1", *(["-DCLOVER_HIP_EXPLICIT_SYNC"] if explicit else []),
                    f"-I{ROOT / 'include'}", str(ROOT / "tests" / "cpp" / "explicit_sync.cpp"), "-o", str(exe), f"-L{lib.parent}", "-lclover_hip",
                    f"-Wl,-rpath,{lib.parent}", "-L/opt/rocm/lib", "-Wl,-rpath,/opt/rocm/lib", "-lpthread"], check=True)
    return exe


@pytest.mark.parametrize("explicit", [True, False])
def test_explicit_sync_client_builds(tmp_path, explicit):
    p = subprocess.run([str(_build_explicit(tmp_path, explicit))], capture_output=True, text=True, timeout=120)
    assert p.returncode == 0 and ("no_device" in p.stdout or "explicit_sync ok" in p.stdout), (p.returncode, p.stdout, p.stderr)
    # the explicit build references neither the handler nor the memory-file machinery at run time: no sigaction / mprotect / memfd calls
    if explicit:
        syms = subprocess.run(["nm", "-C", "--undefined-only", str(tmp_path / "explicit_sync")], capture_output=True, text=True).stdout
        assert "mprotect" not in syms and "pthread_create" not in syms, syms


@pytest.mark.gpu
@pytest.mark.parametrize("explicit", [True, False])
def test_explicit_sync_mode_on_the_gpu(tmp_path, explicit):
    """the same client in both builds: explicit residency (host owns SIGSEGV, pointers re-taken after device operations) and the default
    page-tracked build; device results equal the scalar host twins bit for bit in both"""
    p = subprocess.run([str(_build_explicit(tmp_path, explicit))], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0 and "explicit_sync ok" in p.stdout and f"mode={'explicit' if explicit else 'tracked'}" in p.stdout, (p.returncode, p.stdout, p.stderr)


@pytest.mark.gpu
def test_dropin_example_in_explicit_mode_gives_the_same_answers(tmp_path):
    """the README-style client (tests/cpp/dropin_example.cpp) prints the same lines in both builds"""
    outs = []
    for name, extra in (("dropin_tracked", ()), ("dropin_explicit", ("-DCLOVER_HIP_EXPLICIT_SYNC",))):
        exe = tmp_path / name
        compile_example(exe, extra)
        outs.append(subprocess.run([str(exe)], check=True, capture_output=True, text=True, timeout=300).stdout)
    strip = lambda o: [ln for ln in o.splitlines() if "_us=" not in ln and "_ms=" not in ln]      # noqa: E731 -- timing lines differ
    assert strip(outs[0]) == strip(outs[1])
