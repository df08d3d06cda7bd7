#!/bin/bash
R=$PWD; OUT=$R/gpurun_out/r04b; mkdir -p $OUT
timeout 1500 bash tools/collect_profiles.sh r04 < /dev/null > $OUT/collect.log 2>&1; tail -3 $OUT/collect.log
