#!/usr/bin/env python3
"""Turn a rocprofv3 (ROCm 7.2, rocpd sqlite output) kernel trace into a --stats style summary.

usage: python profiles/summarize_rocpd.py gpurun_out/<dir>/<name>_results.db > profiles/<name>.kernel_stats.txt
"""
import sqlite3
import sys


def main(path):
    con = sqlite3.connect(path)
    rows = con.execute(
        "select name, count(*), sum(duration), avg(duration), min(duration), max(duration), "
        "max(vgpr_count), max(lds_size), max(scratch_size) from kernels group by name "
        "order by sum(duration) desc").fetchall()
    total = sum(r[2] for r in rows) or 1
    print(f"# rocprofv3 --kernel-trace --stats summary of {path}")
    print(f"# total kernel time {total / 1e6:.3f} ms over {sum(r[1] for r in rows)} dispatches")
    print(f"{'calls':>7} {'total_ms':>11} {'avg_us':>11} {'min_us':>10} {'max_us':>10} {'pct':>6} "
          f"{'vgpr':>5} {'lds':>7} {'scratch':>7}  kernel")
    for name, calls, tot, avg, mn, mx, vg, lds, scr in rows:
        print(f"{calls:7d} {tot / 1e6:11.3f} {avg / 1e3:11.2f} {mn / 1e3:10.2f} {mx / 1e3:10.2f} "
              f"{100.0 * tot / total:6.2f} {vg or 0:5d} {lds or 0:7d} {scr or 0:7d}  {name}")


if __name__ == "__main__":
    main(sys.argv[1])
