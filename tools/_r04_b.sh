#!/bin/bash
R=$PWD; OUT=$R/gpurun_out/r04b; mkdir -p $OUT
( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 < /dev/null > $OUT/r04_bench_n1.json 2> $OUT/bench.err ) 2>&1 | grep real
timeout 600 python tools/kernel_bench.py < /dev/null > $OUT/r04_kernel_bench.json 2> $OUT/kb.err; tail -c 200 $OUT/kb.err
