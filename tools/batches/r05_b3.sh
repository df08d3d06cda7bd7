#!/bin/bash
# round 5, GPU batch 3: block-scalar scaleAndAdd kernel -- parity, A/B against the plain kernel, counters
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r05_b3; mkdir -p $O
export TMPDIR=/tmp
( timeout 1500 python -m pytest tests/test_next_rows.py tests/test_gpu_random_shapes.py tests/test_gpu_whole.py tests/test_bench_launch.py tests/test_gpu_parity.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log )
tail -8 $O/pytest.log
for v in default saa_plain; do
  lib=""; [ $v != default ] && lib=tools/_build/variants/libclover_hip_$v.so
  CLV_LIB=$lib KB_ONLY=scale_and_add_n2 timeout 300 python tools/kernel_bench.py > $O/kb_saa_$v.json 2> $O/kb_saa_$v.err
  cat $O/kb_saa_$v.json
done
timeout 200 python tools/dot_fast_ab.py > $O/dot_fast_default.json 2> $O/dot_fast.err; cat $O/dot_fast_default.json
bash tools/weak_kernels_pmc.sh > $O/weak_kernels_pmc.txt 2> $O/weak_kernels_pmc.err
grep "scale_and_add_blk" $O/weak_kernels_pmc.txt
echo "batch done"
