#!/bin/bash
# Round profiles, run on the GPU box from the repository root:   bash tools/collect_profiles.sh r02
# Kernel-trace statistics and counter passes are SEPARATE rocprofv3 runs (gpurun refuses --pmc together with tracing).
# Everything lands under gpurun_out/profiles_$TAG/; copy what should be judged into profiles/.
TAG=${1:-rXX}
R=$PWD
OUT=$R/gpurun_out/profiles_$TAG
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/prof_$TAG; mkdir -p /tmp/prof_$TAG
# 1. the bench command itself (headline mvm, N=1) under --kernel-trace --stats
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_$TAG/bench -o b -- python $R/bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-extras --no-c5 > $OUT/bench_under_rocprof.json 2> /tmp/prof_$TAG/bench.err
python $R/tools/rocpd_summary.py $(find /tmp/prof_$TAG/bench -name "*.db" | head -1) > $OUT/${TAG}_mvm_c3_kernel_stats.txt 2>&1
# 2. the C5 shard (131072 x 65536 per GPU) on this one GPU
timeout 600 python $R/bench.py --preset c5-weak --steps 100 --warmup 10 --no-extras --no-cpu-baseline > $OUT/${TAG}_bench_c5shard_n1.json 2> /tmp/prof_$TAG/c5.err
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_$TAG/c5 -o b -- python $R/bench.py --preset c5-weak --steps 50 --warmup 10 --no-cpu-baseline --no-extras > /dev/null 2>&1
python $R/tools/rocpd_summary.py $(find /tmp/prof_$TAG/c5 -name "*.db" | head -1) > $OUT/${TAG}_mvm_c5shard_kernel_stats.txt 2>&1
# 2b. BASELINE configs[4] whole (2^20 x 2^16, 32 GiB) on this one GPU: what the default line's `c5` object times at N = 1
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_$TAG/c5w -o b -- python $R/bench.py --preset c5-strong --steps 30 --warmup 5 --no-cpu-baseline --no-extras > $OUT/${TAG}_bench_c5whole_under_rocprof_n1.json 2> /tmp/prof_$TAG/c5w.err
python $R/tools/rocpd_summary.py $(find /tmp/prof_$TAG/c5w -name "*.db" | head -1) > $OUT/${TAG}_mvm_c5whole_kernel_stats.txt 2>&1
# 3. GEMM 8192^3 and the exact dot: kernel statistics
GP_CALLS=150 timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_$TAG/gemm -o b -- python $R/tools/gemm_probe.py > /dev/null 2>&1
python $R/tools/rocpd_summary.py $(find /tmp/prof_$TAG/gemm -name "*.db" | head -1) > $OUT/${TAG}_gemm_fp6_8192_kernel_stats.txt 2>&1
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_$TAG/dot -o b -- python $R/tools/dot_probe.py > /dev/null 2>&1
python $R/tools/rocpd_summary.py $(find /tmp/prof_$TAG/dot -name "*.db" | head -1) > $OUT/${TAG}_dot_n2p24_kernel_stats.txt 2>&1
# 4. HBM traffic of the headline kernel: FETCH_SIZE and WRITE_SIZE in separate passes
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 900 rocprofv3 --pmc $c --output-format csv -d /tmp/prof_$TAG/pmc/$c -- python $R/tools/pmc_probe.py > /tmp/prof_$TAG/pmc_$c.log 2>&1
done
python $R/tools/make_pmc_json.py /tmp/prof_$TAG/pmc $OUT/${TAG}_mvm_c3_pmc.json > $OUT/pmc_summary.txt 2>&1
# 5. GEMM counters (four passes)
cd $R
timeout 1200 bash tools/gemm_pmc.sh > $OUT/${TAG}_gemm_fp6_8192_pmc.txt 2>&1
ls -la $OUT
