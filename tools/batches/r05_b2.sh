#!/bin/bash
# round 5, GPU batch 2: dot FAST load-depth A/B, the new bench tests, bench N=1 again, counters of the weak kernels
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r05_b2; mkdir -p $O
export TMPDIR=/tmp
for u in 1 2 4; do CLV_DOT_FAST_U=$u timeout 200 python tools/dot_fast_ab.py >> $O/dot_fast_ab.jsonl 2>> $O/dot_fast_ab.err; done
CLV_DOT_FAST_TWO_LAUNCHES=1 timeout 200 python tools/dot_fast_ab.py >> $O/dot_fast_ab.jsonl 2>> $O/dot_fast_ab.err
timeout 200 python tools/dot_fast_ab.py >> $O/dot_fast_ab.jsonl 2>> $O/dot_fast_ab.err
cat $O/dot_fast_ab.jsonl
( timeout 1500 python -m pytest tests/test_bench_launch.py tests/test_next_rows.py tests/test_sharded_cpp.py tests/test_cpp_dropin.py tests/test_gpu_large.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log )
tail -8 $O/pytest.log
timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_n1.json 2> $O/bench_n1.err
python - <<'PY'
import json
d=json.loads([l for l in open("gpurun_out/r05_b2/bench_n1.json") if l.startswith("{")][-1])
print("lines", sum(1 for l in open("gpurun_out/r05_b2/bench_n1.json")))
print("ranks", d["ms_per_step"], d["value"], "one_process", d["one_process"]["ms_per_step"], d["one_process"]["kernel_avg_ms"], "gemm_sharded", d["gemm_sharded"]["ms_per_step"], d["gemm_sharded"]["per_rank_kernel_ms"])
print({k:(v["frac"] if isinstance(v,dict) and "frac" in v else None) for k,v in d["extras"]["hbm_resident_n2^30"].items()})
PY
bash tools/weak_kernels_pmc.sh > $O/weak_kernels_pmc.txt 2> $O/weak_kernels_pmc.err
tail -40 $O/weak_kernels_pmc.txt
echo "batch done"
