#!/usr/bin/env python
"""HBM bytes per launch and kernel family from two rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE in separate runs, as
MI355X_MICROARCH.md prescribes).  Both counters are in KB; FETCH_SIZE is doubled on gfx950 (tools/pmc_cal.py: a 512 MiB
coalesced read reports 262200 KB), WRITE_SIZE is taken as reported.
    pmc_traffic.py fetch_results.db write_results.db ["bench args" [workload_key]] > profiles/<tag>_pmc_traffic.json"""
import json
import sqlite3
import sys
from collections import defaultdict

# conv_fwd = every forward / dgrad form of the convolution (one-tile conv_fwd_kernel, conv_fwd_v2_kernel, conv_fwd_ws*_kernel
# and conv_thin_kernel): the same launch set bench.py's roofline object prices as the family "conv_fwd"
FAMILIES = [("wgrad_reduce", "wgrad_reduce"), ("conv_wgrad", "conv_wgrad"), ("conv_fwd", "conv_fwd"), ("conv_thin", "conv_fwd"),
            ("bn_act_bwd_apply", "bn_bwd_apply"), ("bn_act_bwd_reduce", "bn_bwd_reduce"), ("bn_fused_fwd", "bn_fwd"),
            ("bn_act_fwd", "bn_fwd"), ("maxpool_bwd", "maxpool_bwd"), ("maxpool_fwd", "maxpool_fwd"),
            ("upsample_bwd", "upsample_bwd"), ("upsample_fwd", "upsample_fwd"), ("pack_weights", "pack_weights"),
            ("adamw_step", "adamw_step")]


def family(name):
    for key, fam in FAMILIES:
        if key in name:
            return fam
    return None


def collect(path, counter):
    c = sqlite3.connect(path)
    q = ("select s.kernel_name, e.value, d.end - d.start from rocpd_pmc_event e join rocpd_info_pmc p on e.pmc_id = p.id "
         "join rocpd_kernel_dispatch d on d.event_id = e.event_id join rocpd_info_kernel_symbol s on d.kernel_id = s.id "
         "where p.name = ?")
    acc = defaultdict(lambda: [0, 0.0, 0.0])
    for name, val, dur in c.execute(q, (counter,)):
        f = family(name)
        if f:
            a = acc[f]
            a[0] += 1
            a[1] += val
            a[2] += dur
    return acc


def main(fetch_db, write_db, bench_args="", key="c3-unet_lc-12x3x512-bf16"):
    rd, wr = collect(fetch_db, "FETCH_SIZE"), collect(write_db, "WRITE_SIZE")
    out = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) --kernel-trace -- python bench.py "
                     + bench_args + "; 1x MI355X (tools/pmc_round.sh)",
           "workload_key": key,
           "corrections": "FETCH_SIZE (KB) doubled on gfx950 (tools/pmc_cal.py calibration, MI355X_MICROARCH.md); "
                          "WRITE_SIZE (KB) as reported",
           "families": {}}
    for fam in rd:
        n, kb, dur = rd[fam]
        nw, kbw, _ = wr.get(fam, [0, 0.0, 0.0])
        r = 2.0 * kb * 1024.0 / n
        w = kbw * 1024.0 / nw if nw else 0.0
        out["families"][fam] = {"launches": n, "hbm_read_bytes_per_launch": int(r), "hbm_write_bytes_per_launch": int(w),
                                "hbm_bytes_per_launch": int(r + w), "avg_us_under_pmc": round(dur / n / 1e3, 2)}
    json.dump(out, sys.stdout, indent=1)


main(*sys.argv[1:])
