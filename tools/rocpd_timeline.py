#!/usr/bin/env python
"""Dispatch-by-dispatch timeline around the idle gaps of a rocprofv3 rocpd database: for the LAST `--tail-ms`, every run of events that
contains a gap >= `--min-gap-us` is printed with start offsets, durations, grid sizes and the gap before each event.
Usage: rocpd_timeline.py results.db [--tail-ms 100] [--min-gap-us 80] [--context 3]"""
import argparse
import sqlite3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("db")
    ap.add_argument("--tail-ms", type=float, default=100.0)
    ap.add_argument("--min-gap-us", type=float, default=80.0)
    ap.add_argument("--context", type=int, default=3)
    ap.add_argument("--max-lines", type=int, default=160)
    a = ap.parse_args()
    c = sqlite3.connect(a.db)
    rows = c.execute("select d.start, d.end, s.kernel_name, d.grid_size_x from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s "
                     "on d.kernel_id = s.id order by d.start").fetchall()
    try:
        cps = c.execute("select start, end, name, size from rocpd_memory_copy order by start").fetchall()
    except sqlite3.Error:
        cps = []
    ev = sorted([(s, e, n[:70], g) for s, e, n, g in rows] + [(s, e, "memcpy:" + str(n), sz) for s, e, n, sz in cps])
    t_end = max(e for _, e, _, _ in ev)
    ev = [x for x in ev if x[0] >= t_end - a.tail_ms * 1e6]
    t0 = ev[0][0]
    cur_e = ev[0][1]
    marks = []
    for i in range(1, len(ev)):
        if ev[i][0] - cur_e >= a.min_gap_us * 1e3:
            marks.append(i)
        cur_e = max(cur_e, ev[i][1])
    show = set()
    for i in marks:
        show.update(range(max(0, i - a.context), min(len(ev), i + a.context + 1)))
    cur_e, last, lines = ev[0][1], -2, 0
    for i, (s, e, n, g) in enumerate(ev):
        gap = (s - cur_e) / 1e3 if i else 0.0
        if i in show and lines < a.max_lines:
            if i != last + 1:
                print("   ...")
            print(f"  t={(s - t0) / 1e3:10.1f} us  gap {gap:8.1f}  dur {(e - s) / 1e3:8.1f}  grid/size {g:>10}  {n}")
            last, lines = i, lines + 1
        cur_e = max(cur_e, e)


if __name__ == "__main__":
    main()
