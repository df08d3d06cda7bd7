#!/usr/bin/env python3
"""Builds profiles/rNN_mvm_c3_pmc.json from rocprofv3 --pmc passes over tools/pmc_probe.py (one counter per pass: FETCH_SIZE, WRITE_SIZE).

    python tools/make_pmc_json.py <dir with the counter_collection.csv files> <out.json>

Units and the gfx950 correction exactly as MI355X_MICROARCH.md prescribes: FETCH_SIZE / WRITE_SIZE count KB; FETCH_SIZE counts 128-byte
read requests as 64 bytes on gfx950 -- calibrated in the same passes on k_read_bw, which reads exactly 2 GiB (16 B per lane, coalesced).
"""
import collections
import csv
import glob
import json
import sys

src, out = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(list)
for f in sorted(glob.glob(src + "/**/*counter_collection.csv", recursive=True)):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0].replace("void ", "")
        agg[(k, row["Counter_Name"])].append(float(row["Counter_Value"]))
kernels = collections.defaultdict(dict)
for (k, c), v in sorted(agg.items()):
    kernels[k][c] = {"n": len(v), "avg": sum(v) / len(v), "min": min(v), "max": max(v)}
cal = kernels.get("k_read_bw", {}).get("FETCH_SIZE", {}).get("avg")
factor = (2 * 1024 * 1024 * 1024) / (cal * 1024) if cal else 2.0          # bytes really read per counted byte
res = {"_how": "rocprofv3 --pmc <one counter> --output-format csv -- python tools/pmc_probe.py; FETCH_SIZE and WRITE_SIZE in separate passes, "
               "averages per dispatch; KB units; gfx950 FETCH_SIZE correction calibrated on k_read_bw (exactly 2 GiB read)",
       "calibration": {"k_read_bw_fetch_size_kb": cal, "bytes_per_counted_byte": factor}, "kernels": kernels}
rows = cols = 65536
alg = rows * cols // 2 + 4 * (rows // 64) * (cols // 64) + (cols // 2 + cols // 16) + (rows // 2 + rows // 16)
for k, c in kernels.items():
    if k.startswith("k_m4_mvm64") and "FETCH_SIZE" in c:
        rd = c["FETCH_SIZE"]["avg"] * 1024 * factor
        wr = c.get("WRITE_SIZE", {}).get("avg", 0.0) * 1024
        res["mvm_c3"] = {"rows": rows, "cols": cols, "kernel": k, "algorithmic_bytes": alg, "hbm_read_bytes": rd, "hbm_write_bytes": wr,
                         "traffic_bytes_per_launch": rd + wr, "traffic_over_algorithmic": (rd + wr) / alg}
        break
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res.get("mvm_c3"), indent=1))
