#!/bin/bash
# round 5, GPU batch 5: the rewritten reference heap walk -- parity and time
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r05_b5; mkdir -p $O
export TMPDIR=/tmp
( timeout 1500 python -m pytest tests/test_next_rows.py tests/test_cpp_dropin.py tests/test_mixed8.py tests/test_iht_recovery.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log )
tail -6 $O/pytest.log
g++ -std=c++11 -O2 -DCLOVER_STOCHASTIC_ROUNDING_DISABLED=1 -Iinclude tools/iht_dropin.cpp -o /tmp/iht_dropin -Lclover_amd/lib -lclover_hip -Wl,-rpath,$PWD/clover_amd/lib -L/opt/rocm/lib -Wl,-rpath,/opt/rocm/lib && timeout 300 /tmp/iht_dropin | tee $O/iht_dropin.json
echo "batch done"
