#!/usr/bin/env python
"""Per-kernel averages of the PMC counters in one or more rocprofv3 rocpd databases.
Usage: rocpd_pmc.py [--match substr] a_results.db [b_results.db ...]"""
import sqlite3
import sys
from collections import defaultdict


def main(argv):
    match = None
    if argv and argv[0] == "--match":
        match, argv = argv[1], argv[2:]
    table = defaultdict(dict)
    for path in argv:
        c = sqlite3.connect(path)
        q = ("select s.kernel_name, p.name, avg(e.value), count(*), avg(d.end - d.start), max(d.grid_size_x), "
             "max(s.arch_vgpr_count), max(d.group_segment_size) "
             "from rocpd_pmc_event e join rocpd_info_pmc p on e.pmc_id = p.id "
             "join rocpd_kernel_dispatch d on d.event_id = e.event_id "
             "join rocpd_info_kernel_symbol s on d.kernel_id = s.id group by s.kernel_name, p.name")
        for kname, pname, val, n, dur, grid, vgpr, lds in c.execute(q):
            if match and match not in kname:
                continue
            table[kname][pname] = val
            table[kname]["_dur_us"] = dur / 1e3
            table[kname]["_calls"] = n
            table[kname]["_grid"] = grid
            table[kname]["_vgpr"] = vgpr
            table[kname]["_lds"] = lds
    for k, d in sorted(table.items(), key=lambda kv: -kv[1].get("_dur_us", 0)):
        print(k[:100])
        print("   " + "  ".join(f"{n}={v:.4g}" for n, v in sorted(d.items())))


if __name__ == "__main__":
    main(sys.argv[1:])
