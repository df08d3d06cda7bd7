cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/thr3 && TP_LOGN=${1:-28} TP_CALLS=5 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/thr3 -- python $GRAFT_REPO_ROOT/tools/thresh_probe.py < /dev/null > /dev/null 2>&1; f=$(find /tmp/thr3 -name "*kernel_stats.csv" | head -1); test -n "$f" && python3 -c "
import csv,sys
for r in csv.DictReader(open('$f')):
    print(f\"{r['Name'][:40]:40s} calls {r['Calls']:>3s} avg {float(r['AverageNs'])/1000:8.2f} us\")
"
