#!/bin/bash
# round 5, GPU batch 6: packed fmas in the stochastic quantiser, graph-replay test of the one-launch dot
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r05_b6; mkdir -p $O
export TMPDIR=/tmp
( timeout 1800 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log )
tail -6 $O/pytest.log
KB_ONLY=stochastic timeout 600 python tools/kernel_bench.py > $O/kb_st.json 2> $O/kb_st.err
python - <<'PY'
import json
d=json.load(open("gpurun_out/r05_b6/kb_st.json"))
for k,v in d.items(): print(f"{k:40s} {v['ms']:9.5f} {v['frac_of_8TBs']:.4f}")
PY
echo "batch done"
