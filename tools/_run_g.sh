set -e
python tools/gen_gemm6_loop256.py clover_amd/csrc/gemm6_loop256.inc experiments > /dev/null
python -c "import __graft_entry__ as g; g.build()" > /dev/null 2>&1
for v in 0 1 2 3 4 5 6 7 8 9 10; do for m in prepared i32; do echo -n "v$v "; CLV_GEMM_LOOP=v$v GB_MODE=$m GB_SIZES=8192 python tools/gemm_bench.py; done; done
