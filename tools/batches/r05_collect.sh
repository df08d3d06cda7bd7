#!/bin/bash
# round 5: the profile set DESIGN.md / profiles/README.md cite (run on the GPU box from the repository root through gpurun)
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r05_final; mkdir -p $O
export TMPDIR=/tmp
T0=$(date +%s)
( time timeout 1200 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/r05_bench_n1.json 2> $O/bench_n1.err ) 2> $O/bench_n1.time
echo "bench default: $(( $(date +%s) - T0 )) s"; tail -3 $O/bench_n1.time
CLOVER_BENCH_DEBUG_ONE_GPU=1 timeout 900 python bench.py --gpus 2 --steps 20 --warmup 5 --no-extras > $O/r05_bench_rehearsal_gpus2.json 2> $O/bench_rehearsal.err
timeout 900 python tools/kernel_bench.py > $O/r05_kernel_bench.json 2> $O/kernel_bench.err
bash tools/collect_profiles.sh r05 > $O/collect.log 2>&1
cp gpurun_out/profiles_r05/* $O/ 2>/dev/null
timeout 900 python tools/gemm_pmc_json.py $O/r05_gemm_pmc.json > $O/gemm_pmc_json.log 2>&1
bash tools/weak_kernels_pmc.sh > $O/r05_weak_kernels_pmc.txt 2> $O/weak_kernels_pmc.err
ls -la $O
echo "collect done in $(( $(date +%s) - T0 )) s"
