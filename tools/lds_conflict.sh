#!/bin/bash
# LDS bank conflicts per kernel (run on the GPU box from the repository root): bash tools/lds_conflict.sh > gpurun_out/lds_conflict.txt
R=$PWD; cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/ldsc
timeout 600 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_BUSY_CYCLES --output-format csv -d /tmp/ldsc -- python $R/tools/lds_conflict_probe.py > /tmp/ldsc_o.txt 2>&1 < /dev/null
tail -2 /tmp/ldsc_o.txt
python3 - <<'PY'
import collections, csv, glob
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("/tmp/ldsc/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0].replace("void ", "")
        agg[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
print(f"{'kernel':52s} {'calls':>5s} {'LDS active':>14s} {'bank conflict':>14s} {'conflict / active':>18s}")
for k, c in sorted(agg.items(), key=lambda kv: -sum(kv[1].get("SQ_LDS_BANK_CONFLICT", [0]))):
    act, bc = c.get("SQ_LDS_IDX_ACTIVE", [0]), c.get("SQ_LDS_BANK_CONFLICT", [0])
    a, b = sum(act) / max(len(act), 1), sum(bc) / max(len(bc), 1)
    if a > 0:
        print(f"{k[:52]:52s} {len(act):5d} {a:14.0f} {b:14.0f} {b / a:18.3f}")
PY
