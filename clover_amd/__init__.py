"""clover_amd -- MI355X (gfx950) backend for Clover's 4-bit hot path.

The product is the C-ABI library ``clover_amd/lib/libclover_hip.so`` (sources in ``clover_amd/csrc``,
interface in ``include/clover_hip.h``) plus the C++ containers in ``include/``.  This Python package is
only the build driver and a thin ctypes binding used by the tests and by ``bench.py``; it contains no
CPU fallback: importing :mod:`clover_amd.lib_binding` without the built library raises.
"""

from .build import build_all, build_hip_library, build_oracle, repo_root  # noqa: F401

__all__ = ["build_all", "build_hip_library", "build_oracle", "repo_root"]
