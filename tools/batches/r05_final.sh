#!/bin/bash
# round 5, final check on the GPU: smoke, the whole GPU suite, the driver's command (N = 1) and the two-rank rehearsal
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r05_final5; mkdir -p $O
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
( timeout 1800 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log )
grep -E "passed|failed|rc=" $O/pytest.log | tail -3
( time timeout 1200 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/r05_bench_n1.json 2> $O/bench_n1.err ) 2> $O/bench_n1.time; tail -3 $O/bench_n1.time
CLOVER_BENCH_DEBUG_ONE_GPU=1 timeout 900 python bench.py --gpus 2 --steps 20 --warmup 5 --no-extras > $O/r05_bench_rehearsal_gpus2.json 2> $O/bench_rehearsal.err
wc -l $O/r05_bench_n1.json $O/r05_bench_rehearsal_gpus2.json
echo "batch done"
