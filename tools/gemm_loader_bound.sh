#!/bin/bash
# Round 6: the bound of wave specialisation in the GEMM main loop (loader waves issue every LDS-DMA request and own the stage barriers, the
# MFMA waves issue only fragment reads, MFMAs and the fold): timing-only variants of the SAME loop in the bench-only probe library
# (tools/_build/libclover_hip_probe.so; results wrong by construction): v0 the product's loop; v1 the LDS-DMA requests gone from the MFMA
# waves' instruction stream (nobody issues them); v4 no stage barriers; v13 = v1 + v4: what loader waves could buy at the very most;
# v5 no fold, v9 the floor (MFMAs + fold only).  8192^3, steady state; `prepared` = the product with both FP6 images cached,
# `i32prepared` = the plain FP6 GEMM (exact int32 sums, no fold).
cd "$(dirname "$0")/.."
for mode in prepared i32prepared; do
    for v in v0 v1 v4 v13 v2 v5 v9 v0; do
        echo "GB_MODE=$mode CLV_GEMM_LOOP=$v: $(GB_LIB=probe GB_MODE=$mode GB_SIZES=8192 CLV_GEMM_LOOP=$v timeout 120 python tools/gemm_bench.py 2>&1 < /dev/null | tail -1)"
    done
done
