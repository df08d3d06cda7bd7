#!/bin/bash
# round 5, GPU batch 4: q/16 conversion everywhere, block-scalar phases in the stochastic kernel -- full GPU suite + kernel_bench + counters
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r05_b4; mkdir -p $O
export TMPDIR=/tmp
( timeout 1800 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log )
tail -8 $O/pytest.log
timeout 900 python tools/kernel_bench.py > $O/kernel_bench.json 2> $O/kernel_bench.err
python - <<'PY'
import json
d=json.load(open("gpurun_out/r05_b4/kernel_bench.json"))
for k,v in d.items(): print(f"{k:40s} {v['ms']:9.5f} {v['frac_of_8TBs']:.4f}")
PY
CLV_LIB=tools/_build/variants/libclover_hip_saa_plain.so KB_ONLY=scale_and_add_n2^30 timeout 300 python tools/kernel_bench.py > $O/kb_saa_plain.json 2> $O/kb_saa_plain.err; cat $O/kb_saa_plain.json
bash tools/weak_kernels_pmc.sh > $O/weak_kernels_pmc.txt 2> $O/weak_kernels_pmc.err
grep -E "INSTS_VALU |GRBM_GUI" $O/weak_kernels_pmc.txt
echo "batch done"
