#!/usr/bin/env python
"""Summarise a rocprofv3 rocpd database (``*_results.db``) the way ``--stats`` does: per-kernel calls,
total / average / min / max duration and share of GPU time.  Usage: rocpd_summary.py results.db > out.csv"""
import sqlite3
import sys


def main(path):
    c = sqlite3.connect(path)
    rows = c.execute(
        "select s.kernel_name, count(*), sum(d.end - d.start), avg(d.end - d.start), min(d.end - d.start), "
        "max(d.end - d.start) from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id = s.id "
        "group by s.kernel_name order by 3 desc").fetchall()
    tot = float(sum(r[2] for r in rows)) or 1.0
    print('"Name","Calls","TotalDurationNs","AverageNs","Percentage","MinNs","MaxNs"')
    for name, calls, total, avg, mn, mx in rows:
        print(f'"{name}",{calls},{int(total)},{avg:.1f},{100.0 * total / tot:.2f},{int(mn)},{int(mx)}')


if __name__ == "__main__":
    main(sys.argv[1])
