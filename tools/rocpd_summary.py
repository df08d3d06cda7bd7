#!/usr/bin/env python3
"""Turns a rocprofv3 rocpd database (*.db) into the per-kernel summary text kept under profiles/.

    python tools/rocpd_summary.py gpurun_out/prof/x_results.db > profiles/rNN_what.txt
"""
import sqlite3
import sys


def main(path: str) -> None:
    cur = sqlite3.connect(path).cursor()
    print(f"# rocprofv3 --kernel-trace --stats summary of {path}")
    print(f"{'calls':>6} {'total_us':>12} {'avg_us':>10} {'min_us':>10} {'max_us':>10} {'pct':>6}  vgpr lds  kernel")
    rows = cur.execute(
        "select name, count(*), sum(duration), avg(duration), min(duration), max(duration), max(vgpr_count), max(lds_size) "
        "from kernels group by name order by sum(duration) desc").fetchall()
    total = sum(r[2] for r in rows) or 1
    for name, n, tot, avg, mn, mx, vgpr, lds in rows:
        print(f"{n:6d} {tot / 1e3:12.3f} {avg / 1e3:10.3f} {mn / 1e3:10.3f} {mx / 1e3:10.3f} {100 * tot / total:6.2f}  {vgpr:4d} {lds:6d} {name}")
    try:
        pmc = cur.execute("select name, counter_name, avg(value), count(*) from counters_collection group by name, counter_name").fetchall()
        if pmc:
            print("\n# counters (average per dispatch)")
            for name, cname, val, n in pmc:
                print(f"{cname:>24} {val:18.1f}  x{n:<4d} {name}")
    except sqlite3.Error:
        pass


if __name__ == "__main__":
    main(sys.argv[1])
