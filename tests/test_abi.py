"""CPU-side checks of the drop-in boundary: the C-ABI library loads without a GPU and exports every
symbol include/clover_hip.h declares; the binding's prototype table covers the same set."""
import ctypes
import re
from pathlib import Path

from clover_amd.build import build_hip_library, repo_root
from clover_amd.lib_binding import SIGNATURES, load_library


def declared_symbols():
    text = (repo_root() / "include" / "clover_hip.h").read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(cl[vm][48]?_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_expected_surface():
    syms = declared_symbols()
    for must in ("clv4_quantize", "clv4_restore", "clv4_dot", "clm4_quantize", "clm4_mvm", "clm4_gemm",
                 "clv_last_error", "clv_malloc", "clv_rng_seed", "clv8_quantize", "clm4_mvm_v8"):
        assert must in syms


def test_library_exports_every_declared_symbol():
    path = build_hip_library()
    lib = ctypes.CDLL(str(path))                      # loads on a machine without a GPU
    missing = [s for s in declared_symbols() if not hasattr(lib, s)]
    assert not missing, f"declared in clover_hip.h but not exported: {missing}"


def test_binding_table_matches_header():
    assert sorted(SIGNATURES) == declared_symbols()
    lib = load_library()
    assert lib.clv_version().decode().startswith("clover_hip")


def test_probe_build_is_not_beside_the_product_and_is_refused():
    """the bench-only probe build (GEMM loop variants that are wrong by construction) lives under tools/_build/, says so in clv_version(),
    and the loader refuses it unless asked"""
    import pytest
    from clover_amd.build import build_probe_library, hip_library_path
    from clover_amd.lib_binding import CloverHipError
    probe = build_probe_library()
    assert probe.parent != hip_library_path().parent
    assert [f.name for f in hip_library_path().parent.glob("*.so")] == ["libclover_hip.so"]
    with pytest.raises(CloverHipError, match="probe build"):
        load_library(probe)
    assert load_library(probe, allow_probe=True).clv_version().decode().startswith("clover_hip_probe")
    assert load_library().clv_version().decode().startswith("clover_hip 0.")


def test_error_codes_without_compute():
    lib = load_library()
    # argument validation happens before any device work: callable on a CPU-only box
    assert lib.clv4_quantize(None, 128, None, None, None, None) == -1
    assert b"null" in lib.clv_last_error()
    assert lib.clm4_mvm(1, 1, 100, 128, 1, 1, 1, 1, None, None) == -1   # rows not a multiple of 64
    assert b"multiple of 64" in lib.clv_last_error()
    assert lib.clm4_gemm(1, 1, 192, 128, 1, 1, 128, 1, None) == -1       # a GEMM operand is a whole matrix: multiples of 128
    assert b"multiples of 128" in lib.clv_last_error()
    n = ctypes.c_int(-1)
    assert lib.clv_device_count(ctypes.byref(n)) == 0 and n.value >= 0


def _build_c_program(tmp_path):
    import subprocess
    lib = build_hip_library()
    exe = tmp_path / "abi_from_c"
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", f"-I{repo_root() / 'include'}", str(repo_root() / "tests" / "c" / "abi_from_c.c"),
                    "-o", str(exe), f"-L{lib.parent}", "-lclover_hip", f"-Wl,-rpath,{lib.parent}", "-L/opt/rocm/lib", "-Wl,-rpath,/opt/rocm/lib"],
                   check=True)
    return exe


def test_header_is_plain_c_and_links_from_c(tmp_path):
    """the drop-in boundary is C: gcc -std=c99 -pedantic compiles a client of include/clover_hip.h and links it to the library"""
    import subprocess
    exe = _build_c_program(tmp_path)
    p = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    assert p.returncode == 0 and ("dot=256.0" in p.stdout or "no_device" in p.stdout), (p.returncode, p.stdout, p.stderr)


import pytest  # noqa: E402


@pytest.mark.gpu
def test_c_client_gets_the_readme_answer_on_the_gpu(tmp_path):
    import subprocess
    p = subprocess.run([str(_build_c_program(tmp_path))], capture_output=True, text=True, timeout=120)
    assert p.returncode == 0 and "dot=256.0" in p.stdout, (p.returncode, p.stdout, p.stderr)
