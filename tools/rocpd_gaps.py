#!/usr/bin/env python
"""Timeline of a rocprofv3 rocpd database: GPU-busy time against the span of the dispatches, the largest idle gaps (with the
kernels on either side) and the busy / idle split of the LAST `--tail-ms` of the trace (the steady state after warm-up).
Usage: rocpd_gaps.py results.db [--tail-ms 200] [--top 15]"""
import argparse
import sqlite3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("db")
    ap.add_argument("--tail-ms", type=float, default=200.0)
    ap.add_argument("--top", type=int, default=15)
    ap.add_argument("--min-gap-us", type=float, default=4.0)
    a = ap.parse_args()
    c = sqlite3.connect(a.db)
    rows = c.execute("select d.start, d.end, s.kernel_name from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s "
                     "on d.kernel_id = s.id order by d.start").fetchall()
    try:
        cps = c.execute("select start, end, name from rocpd_memory_copy order by start").fetchall()
    except sqlite3.Error:
        cps = []
    ev = sorted([(s, e, n) for s, e, n in rows] + [(s, e, "memcpy:" + str(n)) for s, e, n in cps])
    if not ev:
        print("no dispatches")
        return
    t_end = max(e for _, e, _ in ev)
    t0 = t_end - a.tail_ms * 1e6
    ev = [x for x in ev if x[0] >= t0]
    span = (max(e for _, e, _ in ev) - ev[0][0]) / 1e3
    busy, cur_s, cur_e = 0.0, ev[0][0], ev[0][1]
    gaps = []
    prev = ev[0]
    for s, e, n in ev[1:]:
        if s > cur_e:
            busy += cur_e - cur_s
            gaps.append((s - cur_e, prev[2], n))
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
        if e >= prev[1]:
            prev = (s, e, n)
    busy += cur_e - cur_s
    busy /= 1e3
    print(f"last {a.tail_ms:.0f} ms: {len(ev)} dispatches, span {span:.0f} us, busy (union) {busy:.0f} us = {100 * busy / span:.1f} %, "
          f"sum of durations {sum(e - s for s, e, _ in ev) / 1e3:.0f} us")
    big = [g for g in gaps if g[0] >= a.min_gap_us * 1e3]
    print(f"idle gaps >= {a.min_gap_us} us: {len(big)} totalling {sum(g[0] for g in big) / 1e3:.0f} us; all gaps {sum(g[0] for g in gaps) / 1e3:.0f} us")
    agg = {}
    for g, p, n in gaps:
        k = (p[:60], n[:60])
        v = agg.setdefault(k, [0, 0.0])
        v[0] += 1
        v[1] += g / 1e3
    for (p, n), (cnt, tot) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:a.top]:
        print(f"  {tot:9.1f} us in {cnt:5d} gaps (avg {tot / cnt:6.2f})  {p}  ->  {n}")


if __name__ == "__main__":
    main()
