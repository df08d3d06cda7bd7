# rocprofv3 kernel stats of clm4_iht_v8 (1000 iterations, K = 2048): persistent / launch-per-step, deterministic / stochastic rounding.
# Writes gpurun_out/ihtp8/*.csv (run on the GPU box from the repository root).
R=$PWD; mkdir -p $R/gpurun_out/ihtp8; cd /tmp; export TMPDIR=/tmp
for mode in v8 v8st; do for p in 1 0; do
    rm -rf /tmp/prof_$mode$p
    CLV_IHT_PERSISTENT=$p IHT_ITERS=1000 IHT_K=2048 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$mode$p -- python $R/tools/iht_probe.py $mode < /dev/null > /dev/null 2>&1
    f=$(find /tmp/prof_$mode$p -name "*kernel_stats.csv" | head -1)
    test -n "$f" && cp "$f" $R/gpurun_out/ihtp8/${mode}_persistent${p}_kernel_stats.csv && echo "== $mode persistent=$p" && cut -c1-50,230- "$f" | head -6
done; done
