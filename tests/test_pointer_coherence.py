"""The reference's raw-pointer contract over two copies (include/clover_device.h): getData()/getScales() pointers stay valid and
current across device operations, views write through.  CPU: the state machine against a fake device; GPU: the containers."""
import signal
import subprocess

import pytest

from clover_amd.build import build_hip_library, repo_root

ROOT = repo_root()
CPP = ROOT / "tests" / "cpp"


def _link_flags():
    lib = build_hip_library()
    return [f"-L{lib.parent}", "-lclover_hip", f"-Wl,-rpath,{lib.parent}", "-L/opt/rocm/lib", "-Wl,-rpath,/opt/rocm/lib"]


def _build_coherence(tmp_path, extra=()):
    exe = tmp_path / "pointer_coherence"
    subprocess.run(["g++", "-std=c++11", "-O2", "-Wall", "-Wextra", "-DCLOVER_STOCHASTIC_ROUNDING_DISABLED=1", f"-I{ROOT / 'include'}", *extra,
                    str(CPP / "pointer_coherence.cpp"), "-o", str(exe), *_link_flags()], check=True)
    return exe


@pytest.mark.parametrize("opt", ["-O0", "-O2", "-O3"])
def test_mirror_state_machine_with_fake_device(tmp_path, opt):
    obj, exe = tmp_path / "fake_clv.o", tmp_path / "mirror_states"
    subprocess.run(["gcc", "-std=c99", "-Wall", "-c", f"-I{ROOT / 'include'}", str(CPP / "fake_clv.c"), "-o", str(obj)], check=True)
    subprocess.run(["g++", "-std=c++11", opt, "-Wall", "-Wextra", f"-I{ROOT / 'include'}", str(CPP / "mirror_states.cpp"), str(obj), "-o", str(exe), "-lpthread"],
                   check=True)
    p = subprocess.run([str(exe)], capture_output=True, text=True, timeout=60)
    assert p.returncode == 0 and "mirror ok" in p.stdout, (p.returncode, p.stdout, p.stderr)


def test_mprotect_enomem_degrades_to_explicit_residency(tmp_path):
    """a refused mprotect (vm.max_map_count) must not end the process: the block falls back to the explicit-residency rules"""
    obj, exe = tmp_path / "fake_clv.o", tmp_path / "mirror_enomem"
    subprocess.run(["gcc", "-std=c99", "-Wall", "-c", f"-I{ROOT / 'include'}", str(CPP / "fake_clv.c"), "-o", str(obj)], check=True)
    subprocess.run(["g++", "-std=c++11", "-O2", "-Wall", "-Wextra", "-DCLOVER_HIP_TEST_MPROTECT_ENOMEM", f"-I{ROOT / 'include'}",
                    str(CPP / "mirror_enomem.cpp"), str(obj), "-o", str(exe), "-lpthread"], check=True)
    p = subprocess.run([str(exe)], capture_output=True, text=True, timeout=60)
    assert p.returncode == 0 and "mirror enomem ok" in p.stdout and "falls back to explicit residency" in p.stderr, (p.returncode, p.stdout, p.stderr)


def _build_mirror_threads(tmp_path, *flags):
    obj, exe = tmp_path / "fake_clv.o", tmp_path / "mirror_threads"
    subprocess.run(["gcc", "-std=c99", "-Wall", "-c", f"-I{ROOT / 'include'}", str(CPP / "fake_clv.c"), "-o", str(obj)], check=True)
    subprocess.run(["g++", "-std=c++11", *flags, "-Wall", "-Wextra", f"-I{ROOT / 'include'}", str(CPP / "mirror_threads.cpp"), str(obj), "-o", str(exe),
                    "-lpthread"], check=True)
    return exe


@pytest.mark.parametrize("flags", [("-O2",), ("-O0",), ("-O2", "-DCLOVER_HIP_FAULT_INLINE"), ("-O1", "-g", "-fsanitize=thread")],
                         ids=["O2", "O0", "inline-handler", "tsan"])
def test_mirror_under_threads_with_fake_device(tmp_path, flags):
    """four threads faulting on one block, writers racing uploads, the block table changing under lookups, a destructor waiting for
    a fault in flight (tests/cpp/mirror_threads.cpp); also with the copy made inside the handler, and under ThreadSanitizer"""
    exe = _build_mirror_threads(tmp_path, *flags)
    for _ in range(2):
        p = subprocess.run([str(exe)], capture_output=True, text=True, timeout=300)
        assert p.returncode == 0 and "mirror threads ok" in p.stdout and "WARNING: ThreadSanitizer" not in p.stderr, (p.returncode, p.stdout, p.stderr[-3000:])


def _build_pointer_threads(tmp_path):
    exe = tmp_path / "pointer_threads"
    subprocess.run(["g++", "-std=c++11", "-O2", "-Wall", "-Wextra", "-DCLOVER_STOCHASTIC_ROUNDING_DISABLED=1", f"-I{ROOT / 'include'}",
                    str(CPP / "pointer_threads.cpp"), "-o", str(exe), *_link_flags(), "-lpthread"], check=True)
    return exe


def test_pointer_threads_client_builds(tmp_path):
    p = subprocess.run([str(_build_pointer_threads(tmp_path))], capture_output=True, text=True, timeout=120)
    assert p.returncode == 0 and ("no_device" in p.stdout or "pointer threads ok" in p.stdout), (p.returncode, p.stdout, p.stderr)


@pytest.mark.gpu
def test_kept_pointers_under_host_threads_on_the_gpu(tmp_path):
    """mvm / quantize / restore from several threads on shared operands while other threads read and write through kept pointers"""
    p = subprocess.run([str(_build_pointer_threads(tmp_path))], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0 and "pointer threads ok" in p.stdout, (p.returncode, p.stdout, p.stderr[-2000:])


def test_coherence_client_builds_and_a_wild_access_still_crashes(tmp_path):
    exe = _build_coherence(tmp_path, extra=("-mavx2",))          # -mavx2: the setRandomKeys(__m256i, __m256i) overloads compile
    p = subprocess.run([str(exe)], capture_output=True, text=True, timeout=60)
    assert p.returncode == 0 and ("no_device" in p.stdout or "coherence ok" in p.stdout), (p.returncode, p.stdout, p.stderr)
    p = subprocess.run([str(exe), "crash"], capture_output=True, text=True, timeout=60)
    assert p.returncode == -signal.SIGSEGV and "not reached" not in p.stdout, (p.returncode, p.stdout)


def test_m256_key_overload_exists_only_with_avx(tmp_path):
    src = tmp_path / "keys.cpp"
    src.write_text('#include "CloverMatrix4.h"\n#include <immintrin.h>\nvoid f(CloverVector4 &v, CloverVector8 &w, CloverMatrix4 &m, __m256i a, __m256i b)'
                   " { v.setRandomKeys(a, b); w.setRandomKeys(a, b); m.setRandomKeys(a, b); }\n")
    subprocess.run(["g++", "-std=c++11", "-mavx2", "-fsyntax-only", f"-I{ROOT / 'include'}", str(src)], check=True)


@pytest.mark.gpu
def test_kept_pointers_and_views_behave_like_the_reference(tmp_path):
    exe = _build_coherence(tmp_path)
    p = subprocess.run([str(exe)], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0 and "coherence ok" in p.stdout, (p.returncode, p.stdout, p.stderr)
    us = float(p.stdout.split("c1_header_dot_us=")[1].split()[0])
    print(f"C1 (n=128) dot through the C++ headers: {us:.1f} us per call")
    assert us < 200.0


def _build_validate(tmp_path, name="validate_relations", extra=()):
    exe = tmp_path / (name + ("_x" if extra else ""))
    subprocess.run(["g++", "-std=c++11", "-O2", "-Wall", "-Wextra", "-DCLOVER_STOCHASTIC_ROUNDING_DISABLED=1", *extra, f"-I{ROOT / 'include'}",
                    str(CPP / f"{name}.cpp"), "-o", str(exe), *_link_flags()], check=True)
    return exe


def test_validation_relations_client_builds(tmp_path):
    p = subprocess.run([str(_build_validate(tmp_path))], capture_output=True, text=True, timeout=120)
    assert p.returncode == 0 and ("no_device" in p.stdout or "validate ok" in p.stdout), (p.returncode, p.stdout[-2000:], p.stderr)


@pytest.mark.gpu
def test_device_methods_against_their_scalar_twins_like_the_reference_harness(tmp_path):
    """test/validate/02_vector.cpp + 03_matrix.cpp: kernel vs `_scalar` host twin (include/clover_scalar.h), exact where the reference is exact"""
    p = subprocess.run([str(_build_validate(tmp_path))], capture_output=True, text=True, timeout=900)
    assert p.returncode == 0 and "validate ok" in p.stdout, (p.returncode, p.stdout[-3000:], p.stderr[-1000:])


def test_validation_grid_client_builds(tmp_path):
    p = subprocess.run([str(_build_validate(tmp_path, "validate_grid"))], capture_output=True, text=True, timeout=120)
    assert p.returncode == 0 and ("no_device" in p.stdout or "validate grid ok" in p.stdout), (p.returncode, p.stdout[-2000:], p.stderr)


@pytest.mark.gpu
def test_reference_validation_grid_at_full_density(tmp_path):
    """the reference's own grid, not a sample of it: every n = 128 ... 2047 (test/validate/02_vector.cpp:111-553) and every
    (128 i) x (128 j), i, j = 1 ... 10 (03_matrix.cpp:38-573), same relations and strictness"""
    p = subprocess.run([str(_build_validate(tmp_path, "validate_grid"))], capture_output=True, text=True, timeout=1500)
    assert p.returncode == 0 and "validate grid ok" in p.stdout, (p.returncode, p.stdout[-3000:], p.stderr[-1000:])
    print(p.stdout.strip().splitlines()[-2])


@pytest.mark.gpu
def test_reference_validation_grid_in_the_explicit_residency_build(tmp_path):
    """the same full-density grid with the containers built -DCLOVER_HIP_EXPLICIT_SYNC (no signal handler, no mprotect): every relation
    is written with accessors or with pointers taken after the device operation, which is that build's one rule"""
    p = subprocess.run([str(_build_validate(tmp_path, "validate_grid", extra=("-DCLOVER_HIP_EXPLICIT_SYNC",)))], capture_output=True, text=True,
                       timeout=1500)
    assert p.returncode == 0 and "validate grid ok" in p.stdout, (p.returncode, p.stdout[-3000:], p.stderr[-1000:])


@pytest.mark.gpu
@pytest.mark.parametrize("build", ["tracked", "explicit"])
def test_random_operation_sequences_against_a_host_model(tmp_path, build):
    """tests/cpp/mirror_fuzz.cpp: device kernels, accessors, raw writes and reads through kept getData() / getScales() pointers, copies and a
    view, in random order -- one object compared with a host-only model after every step.  Ten seeds x 400 steps per container build."""
    exe = tmp_path / "mirror_fuzz"
    extra = ["-DCLOVER_HIP_EXPLICIT_SYNC"] if build == "explicit" else []
    subprocess.run(["g++", "-std=c++11", "-O2", "-Wall", "-Wextra", "-DCLOVER_STOCHASTIC_ROUNDING_DISABLED=1", f"-I{ROOT / 'include'}", *extra,
                    str(CPP / "mirror_fuzz.cpp"), "-o", str(exe), *_link_flags(), "-lpthread"], check=True)
    for seed in range(10):
        p = subprocess.run([str(exe), str(1000 + seed), "400"], capture_output=True, text=True, timeout=120)
        assert p.returncode == 0 and p.stdout.strip().startswith("ok"), (seed, p.stdout[-800:], p.stderr[-400:])
