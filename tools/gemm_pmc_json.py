#!/usr/bin/env python3
"""GEMM counters for bench.py's `gemm.roofline` (traffic, mfma_busy_pct): run ON THE GPU BOX from the repository root,

    python tools/gemm_pmc_json.py gpurun_out/r03_gemm_pmc.json        # then copy the file into profiles/

Separate rocprofv3 --pmc passes (gpurun refuses counters together with tracing; FETCH_SIZE and WRITE_SIZE cannot share a pass) over
a probe that launches a calibration read of exactly 2 GiB (k_read_bw) and the 8192^3 clm4_gemm 20 times.  Units and the gfx950
correction as MI355X_MICROARCH.md prescribes: FETCH_SIZE / WRITE_SIZE count KB, FETCH_SIZE counts 128-byte requests as 64 B on gfx950 --
calibrated in the same pass.  A clm4_gemm call = k_m4_to_fp6 (re-coding) + k_m4_gemm_fp6_t256: both kernels' bytes are added."""
import collections
import csv
import glob
import json
import os
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
G = 8192

if len(sys.argv) > 1 and sys.argv[1] == "--probe":
    import ctypes as C
    sys.path.insert(0, str(ROOT))
    from clover_amd.build import build_probe_library
    from clover_amd.lib_binding import CloverHip
    hip = CloverHip(path=build_probe_library(), allow_probe=True)      # clvx_* live in the bench-only probe build
    lib = hip.lib
    lib.clvx_read_bw.argtypes = [C.c_void_p, C.c_uint64, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    big = hip.alloc(2 << 30)
    out = hip.alloc(256)
    hip.check(lib.clv_fill_random_nibbles(big.ptr, big.nbytes, 1, 0, None))
    for _ in range(3):
        hip.check(lib.clvx_read_bw(big.ptr, big.nbytes, 1, 16, out.ptr, None))
    A, B = hip.alloc(G * G // 2), hip.alloc(G * G // 2)
    sA, sB = hip.alloc((G // 64) ** 2 * 4), hip.alloc((G // 64) ** 2 * 4)
    Cc = hip.alloc(G * G * 4)
    for t, sd in ((A, 21), (B, 22)):
        hip.check(lib.clv_fill_random_nibbles(t.ptr, t.nbytes, sd, 0, None))
    for t, sd in ((sA, 23), (sB, 24)):
        hip.check(lib.clv_fill_random_scales(t.ptr, t.nbytes // 4, sd, 0, None))
    for _ in range(20):
        hip.check(lib.clm4_gemm(A.ptr, sA.ptr, G, G, B.ptr, sB.ptr, G, Cc.ptr, None))
    hip.sync()
    print("gemm pmc probe done")
    sys.exit(0)

out_path = sys.argv[1]
work = "/tmp/gemm_pmc_json"
subprocess.run(["rm", "-rf", work])
env = dict(os.environ, TMPDIR="/tmp")
passes = [["FETCH_SIZE"], ["WRITE_SIZE"], ["SQ_VALU_MFMA_BUSY_CYCLES", "GRBM_GUI_ACTIVE", "SQ_BUSY_CYCLES", "SQ_WAVES"], ["TCC_HIT_sum", "TCC_MISS_sum"]]
for i, p in enumerate(passes):
    subprocess.run(["rocprofv3", "--pmc", *p, "--output-format", "csv", "-d", f"{work}/p{i}", "--", sys.executable, __file__, "--probe"],
                   cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=900)
agg = collections.defaultdict(list)
for f in sorted(glob.glob(work + "/**/*counter_collection.csv", recursive=True)):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0].replace("void ", "").split("<")[0]
        agg[(k, row["Counter_Name"])].append(float(row["Counter_Value"]))
avg = {kc: sum(v) / len(v) for kc, v in agg.items()}
cal = avg.get(("k_read_bw", "FETCH_SIZE"))
factor = (2 << 30) / (cal * 1024) if cal else 2.0
gk, rk = "k_m4_gemm_fp6_t256", "k_m4_to_fp6"
rd = sum(avg.get((k, "FETCH_SIZE"), 0.0) for k in (gk, rk)) * 1024 * factor
wr = sum(avg.get((k, "WRITE_SIZE"), 0.0) for k in (gk, rk)) * 1024
alg = 2 * (G * G // 2 + 4 * (G // 64) ** 2) + 4 * G * G
busy, active = avg.get((gk, "SQ_VALU_MFMA_BUSY_CYCLES")), avg.get((gk, "GRBM_GUI_ACTIVE"))
hit, miss = avg.get((gk, "TCC_HIT_sum")), avg.get((gk, "TCC_MISS_sum"))
res = {
    "G": G, "kernels": [rk, gk], "calls_averaged": len(agg.get((gk, "FETCH_SIZE"), [])),
    "algorithmic_bytes_per_call": alg,
    "hbm_read_bytes_per_call": round(rd), "hbm_write_bytes_per_call": round(wr), "traffic_bytes_per_call": round(rd + wr),
    "traffic_over_algorithmic": round((rd + wr) / alg, 4),
    "traffic_how": "FETCH_SIZE (KB, x gfx950 factor calibrated on a 2 GiB k_read_bw in the same pass) + WRITE_SIZE (KB), separate rocprofv3 "
                   "--pmc passes, re-coding kernel + matrix kernel, average per clm4_gemm call",
    "fetch_size_calibration_bytes_per_counted_byte": round(factor, 4),
    # SQ_VALU_MFMA_BUSY_CYCLES sums the busy cycles of all 1024 SIMDs; GRBM_GUI_ACTIVE sums the 8 XCDs' active cycles
    "mfma_busy_pct": round(100.0 * busy / (active / 8 * 1024), 2) if busy and active else None,
    "mfma_busy_how": "SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs), matrix kernel only, counter pass (cold clocks)",
    "l2_hit_pct": round(100.0 * hit / (hit + miss), 2) if hit and miss else None,
    "l2_miss_bytes": round(miss * 128) if miss else None,
    "raw": {f"{k}:{c}": v for (k, c), v in sorted(avg.items())},
}
Path(out_path).parent.mkdir(parents=True, exist_ok=True)
json.dump(res, open(out_path, "w"), indent=1)
print(json.dumps({k: v for k, v in res.items() if k != "raw"}, indent=1))
