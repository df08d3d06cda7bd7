#!/usr/bin/env python3
"""Build a VARIANT of libclover_hip.so for same-box A/B runs: the product's objects with ONE source recompiled under extra -D flags.

    python tools/build_variant.py <name> <source.hip> -DFOO=1 [-DBAR=2 ...]   ->  tools/_build/variants/libclover_hip_<name>.so

The variant loads like the product (clv_version says "clover_hip ..."): point tools/kernel_bench.py at it with CLV_LIB=<path>.  Nothing in
the product or the tests ever loads these."""
import subprocess
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from clover_amd.build import HIP_FLAGS, HIP_SOURCES, _hipcc, build_hip_library, hip_library_path, repo_root  # noqa: E402

name, source, flags = sys.argv[1], sys.argv[2], sys.argv[3:]
root = repo_root()
build_hip_library()
obj_dir = hip_library_path().parent / "obj"
out_dir = root / "tools" / "_build" / "variants"
out_dir.mkdir(parents=True, exist_ok=True)
src_dir = root / "clover_amd" / "csrc"
obj = out_dir / f"{Path(source).stem}_{name}.o"
subprocess.run([_hipcc(), *HIP_FLAGS, *flags, f"-I{root / 'include'}", f"-I{src_dir}", "-c", "-o", str(obj), str(src_dir / source)], check=True)
objs = [str(obj_dir / (Path(s).stem + ".o")) for s in HIP_SOURCES if s != source] + [str(obj)]
out = out_dir / f"libclover_hip_{name}.so"
subprocess.run([_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", str(out), *objs, "-ldl"], check=True)
print(out)
