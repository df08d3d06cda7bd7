#!/bin/bash
# Round 4: the bound of the one GEMM lever left (128 x 64 wave tiles at two waves per SIMD), from timing-only variants of the SAME
# main loop in the bench-only probe library (tools/_build/libclover_hip_probe.so; results wrong by construction):
#   v0 the product's loop; v2 no fragment reads at all; v6 no store of C; v11 a quarter of the fragment reads left out (what the
#   larger wave tile saves per MFMA, at UNCHANGED occupancy); v12 = v11 + no store.  8192^3, both operands prepared, steady state.
cd "$(dirname "$0")/.."
for v in v0 v11 v12 v6 v2 v9 v0; do
    echo "CLV_GEMM_LOOP=$v: $(GB_LIB=probe GB_MODE=prepared GB_SIZES=8192 CLV_GEMM_LOOP=$v python tools/gemm_bench.py 2>&1 | tail -1)"
done
