#!/bin/bash
# round 5, GPU batch 1: correctness of the round's new code + same-box A/B of the kernel variants (tools/build_variant.py)
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r05_b1; mkdir -p $O
export TMPDIR=/tmp
V=tools/_build/variants
( timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log ) 
tail -5 $O/pytest.log
for v in default saa_u1 saa_u4; do
  lib=""; [ $v != default ] && lib=$V/libclover_hip_$v.so
  CLV_LIB=$lib KB_ONLY=scale_and_add_n2^30 timeout 300 python tools/kernel_bench.py > $O/kb_saa_$v.json 2> $O/kb_saa_$v.err
done
for v in default mvf_8k_w4_u2 mvf_8k_w4_u4 mvf_16k_fence mvf_8k_w3; do
  lib=""; [ $v != default ] && lib=$V/libclover_hip_$v.so
  CLV_LIB=$lib KB_ONLY=mvm_f32 timeout 300 python tools/kernel_bench.py > $O/kb_mvf_$v.json 2> $O/kb_mvf_$v.err
done
KB_ONLY=dot_fast timeout 300 python tools/kernel_bench.py > $O/kb_dot_one_launch.json 2> $O/kb_dot_one.err
CLV_DOT_FAST_TWO_LAUNCHES=1 KB_ONLY=dot_fast timeout 300 python tools/kernel_bench.py > $O/kb_dot_two_launches.json 2> $O/kb_dot_two.err
MB_SIZES=8192x8192,16384x16384,4096x8192 MB_VARIANTS=1,5,7,8 MB_SKIP_READ=1 timeout 300 python tools/microbench.py > $O/microbench_L8.json 2> $O/microbench_L8.err
for f in "" "-DCLOVER_HIP_EXPLICIT_SYNC"; do
  g++ -std=c++11 -O2 -DCLOVER_STOCHASTIC_ROUNDING_DISABLED=1 $f -Iinclude tools/iht_dropin.cpp -o /tmp/iht_dropin -Lclover_amd/lib -lclover_hip -Wl,-rpath,$PWD/clover_amd/lib -L/opt/rocm/lib -Wl,-rpath,/opt/rocm/lib && timeout 300 /tmp/iht_dropin >> $O/iht_dropin.json 2>> $O/iht_dropin.err
done
CLOVER_BENCH_DEBUG_ONE_GPU=1 CLOVER_BENCH_C5_ROWS=16384 timeout 600 python bench.py --gpus 2 --steps 6 --warmup 2 --rows-per-gpu 4096 --cols 8192 --no-extras --cpu-sample-rows 1024 > $O/bench_rehearsal2.json 2> $O/bench_rehearsal2.err
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_n1.json 2> $O/bench_n1.err
echo "batch done"; ls -la $O
for f in $O/kb_*.json $O/microbench_L8.json $O/iht_dropin.json; do echo "== $f"; cat $f; done 2>/dev/null | head -150
