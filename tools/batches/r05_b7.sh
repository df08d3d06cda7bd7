#!/bin/bash
# round 5, GPU batch 7: CloverVector8::dot -- parity (C ABI, headers, validation grid) and time
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r05_b7; mkdir -p $O
export TMPDIR=/tmp
( timeout 1800 python -m pytest tests/test_mixed8.py tests/test_cpp_dropin.py tests/test_abi.py tests/test_gpu_parity.py tests/test_graph_capture.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log )
tail -15 $O/pytest.log
KB_ONLY=dot timeout 600 python tools/kernel_bench.py > $O/kb_dot.json 2> $O/kb_dot.err
python - <<'PY'
import json
d=json.load(open("gpurun_out/r05_b7/kb_dot.json"))
for k,v in d.items(): print(f"{k:40s} {v['ms']:9.5f} {v['frac_of_8TBs']:.4f}")
PY
echo "batch done"
