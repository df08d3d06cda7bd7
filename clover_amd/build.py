"""Build driver: compiles the HIP library (gfx950) and the CPU oracle in-tree.

hipcc cross-compiles for gfx950 without a GPU, so this runs in the CPU-only builder container; the
resulting .so files travel to the GPU box with the repository snapshot.
"""
from __future__ import annotations

import os
import shutil
import subprocess
from pathlib import Path

HIP_SOURCES = ["runtime.hip", "vector4.hip", "matrix4.hip", "rng4.hip", "gemm4.hip", "gemm6.hip", "multi.hip", "scale_add4.hip", "transpose4.hip", "threshold4.hip", "iht4.hip", "iht_persist.hip", "mvm_f32.hip", "mixed8.hip"]
# per-file additions.  gemm6.hip: hipcc's SLP pass pairs the fold's scalar fmas into v_pk_fma_f32, which is slower beside MFMAs
# (MI355X_MICROARCH.md, "price of one filler beside MFMAs"); measured here: see DESIGN.md 6
EXTRA_FLAGS = {}
HIP_FLAGS = [
    "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC",
    # the reference's arithmetic is a fixed sequence of separately rounded fp32 ops + explicit fmas:
    "-ffp-contract=off", "-fhip-fp32-correctly-rounded-divide-sqrt", "-fno-fast-math",
]


def repo_root() -> Path:
    return Path(__file__).resolve().parent.parent


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and Path(cand).exists():
            return cand
    raise RuntimeError("hipcc not found (need ROCm to build libclover_hip.so)")


def _stale(target: Path, deps: list[Path]) -> bool:
    if not target.exists():
        return True
    t = target.stat().st_mtime
    return any(d.stat().st_mtime > t for d in deps if d.exists())


def hip_library_path() -> Path:
    return repo_root() / "clover_amd" / "lib" / "libclover_hip.so"


def build_hip_library(force: bool = False, verbose: bool = False) -> Path:
    root = repo_root()
    src_dir = root / "clover_amd" / "csrc"
    srcs = [src_dir / s for s in HIP_SOURCES if (src_dir / s).exists()]
    deps = srcs + list(src_dir.glob("*.h")) + list(src_dir.glob("*.inc")) + [root / "include" / "clover_hip.h"]
    out = hip_library_path()
    out.parent.mkdir(parents=True, exist_ok=True)
    if force or _stale(out, deps):
        # one object per source (in parallel, with its own flags), then one link
        obj_dir = out.parent / "obj"
        obj_dir.mkdir(exist_ok=True)
        jobs = []
        for src in srcs:
            obj = obj_dir / (src.stem + ".o")
            cmd = [_hipcc(), *HIP_FLAGS, *EXTRA_FLAGS.get(src.name, []), f"-I{root / 'include'}", f"-I{src_dir}", "-c", "-o", str(obj), str(src)]
            if verbose:
                print(" ".join(cmd))
            jobs.append((obj, cmd, subprocess.Popen(cmd)))
        for obj, cmd, proc in jobs:
            if proc.wait() != 0:
                raise subprocess.CalledProcessError(proc.returncode, cmd)
        cmd = [_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", str(out), *[str(o) for o, _, _ in jobs]]
        cmd += ["-ldl"]          # RCCL is dlopen'ed lazily by multi.hip
        if verbose:
            print(" ".join(cmd))
        subprocess.run(cmd, check=True)
    return out


def probe_library_path() -> Path:
    # NOT beside the product library: a mis-set -L / CLV_LIB must never pick up a build whose GEMM variants are wrong by construction
    return repo_root() / "tools" / "_build" / "libclover_hip_probe.so"


def build_probe_library(force: bool = False) -> Path:
    """libclover_hip_probe.so = the product's objects with gemm6.hip compiled a second time under -DCLV_GEMM_EXPERIMENTS: the GEMM main
    loop's timing-only variants with parts left out (tools/gen_gemm6_loop256.py ... experiments; results wrong by construction), selected
    by CLV_GEMM_LOOP=vN -- and matrix4.hip under -DCLV_EXPERIMENTS: the mvm kernel variants and the plain read-bandwidth kernel behind
    tools/microbench.py (clvx_mvm_variant, clvx_read_bw).  It lives under tools/_build/, and clv_version() of it reads "clover_hip_probe ..." (load_library refuses that unless
    allow_probe=True).  BENCH INFRASTRUCTURE: bench.py's `gemm.ceiling` and tools/gemm_bench.py load it explicitly to measure what the
    arithmetic alone costs on the box the bench runs on; nothing else ever loads it and the product library has no such switch."""
    import sys
    root = repo_root()
    src_dir = root / "clover_amd" / "csrc"
    out = probe_library_path()
    out.parent.mkdir(parents=True, exist_ok=True)
    stale_old = repo_root() / "clover_amd" / "lib" / "libclover_hip_probe.so"      # where rounds 2-3 put it
    if stale_old.exists():
        stale_old.unlink()
    build_hip_library(force=force)
    obj_dir = hip_library_path().parent / "obj"
    gen = root / "tools" / "gen_gemm6_loop256.py"
    deps = [src_dir / "gemm6.hip", src_dir / "matrix4.hip", gen, hip_library_path()] + list(src_dir.glob("*.h")) + list(src_dir.glob("*.inc"))
    if force or _stale(out, deps):
        subprocess.run([sys.executable, str(gen), str(obj_dir / "gemm6_loop256_exp.inc"), "experiments"], check=True, stdout=subprocess.DEVNULL)
        obj = obj_dir / "gemm6_probe.o"
        subprocess.run([_hipcc(), *HIP_FLAGS, "-DCLV_GEMM_EXPERIMENTS", f"-I{root / 'include'}", f"-I{src_dir}", f"-I{obj_dir}", "-c", "-o", str(obj),
                        str(src_dir / "gemm6.hip")], check=True)
        # matrix4.hip with its experiment entry points (clvx_read_bw, clvx_mvm_variant: tools/microbench.py, pmc_probe.py) -- the product
        # library is compiled without them
        obj_m = obj_dir / "matrix4_probe.o"
        subprocess.run([_hipcc(), *HIP_FLAGS, "-DCLV_EXPERIMENTS", f"-I{root / 'include'}", f"-I{src_dir}", "-c", "-o", str(obj_m),
                        str(src_dir / "matrix4.hip")], check=True)
        objs = [str(obj_dir / (Path(s).stem + ".o")) for s in HIP_SOURCES if s not in ("gemm6.hip", "matrix4.hip")] + [str(obj), str(obj_m)]
        subprocess.run([_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", str(out), *objs, "-ldl"], check=True)
    return out


def build_oracle(force: bool = False) -> Path:
    """Builds oracle/liboracle.so (+ liboracle_fast.so): test infrastructure, never loaded by the product."""
    odir = repo_root() / "oracle"
    if force:
        subprocess.run(["make", "-C", str(odir), "clean"], check=True, stdout=subprocess.DEVNULL)
    subprocess.run(["make", "-C", str(odir), "all"], check=True, stdout=subprocess.DEVNULL)
    # oracle/_ref: the reference's generator header compiled from /root/reference where that tree exists (this container);
    # a no-op on the GPU box, which uses the prebuilt oracle/_ref/*.so that travelled with the snapshot
    subprocess.run(["make", "-C", str(odir), "ref"], check=True, stdout=subprocess.DEVNULL)
    return odir / "liboracle.so"


def build_all(force: bool = False, verbose: bool = False) -> None:
    build_hip_library(force=force, verbose=verbose)
    build_probe_library(force=force)
    build_oracle(force=force)


if __name__ == "__main__":
    build_all(force=True, verbose=True)
