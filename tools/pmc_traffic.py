#!/usr/bin/env python
"""HBM bytes per launch and kernel family from two rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE in separate runs, as
MI355X_MICROARCH.md prescribes).  Both counters are in KB; FETCH_SIZE is doubled on gfx950 (tools/pmc_cal.py: a 512 MiB
coalesced read reports 262200 KB), WRITE_SIZE is taken as reported.
    pmc_traffic.py fetch_results.db write_results.db ["bench args" [workload_key [mfma_results.db [bench_stdout.log]]]] > profiles/<tag>_pmc_traffic.json
`bench_stdout.log` = the stdout of the counted process (`bench.py --roofline-only` prints roofline.process_totals: launches and
ALGORITHMIC bytes per family over every launch of the process): with it each family also gets hbm_bytes_total,
algorithmic_bytes_total and traffic_over_algorithmic = the ratio of the two TOTALS over the same launch population (round 5;
the per-launch means of a counter pass and of bench.py's 10 instrumented iterations are different head : body mixes).
A third pass (SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_BF16 in ONE run) adds the matrix
pipe's busy share per family: mfma_busy_pct = 100 x sum(SQ_VALU_MFMA_BUSY_CYCLES) / (4 SIMDs x sum(SQ_BUSY_CU_CYCLES-equivalent))
-- see `mfma()` for the normalisation actually used on this stack.  The json records the kernel-source hash of the tree it
was taken in (fedicra_amd._lib.source_hash): bench.py only reports traffic from a file whose hash is the running tree's."""
import json
import os
import sqlite3
import sys
from collections import defaultdict

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

# conv_fwd = every forward / dgrad form of the convolution (one-tile conv_fwd_kernel, conv_fwd_v2_kernel, conv_fwd_ws*_kernel
# and conv_thin_kernel): the same launch set bench.py's roofline object prices as the family "conv_fwd"
FAMILIES = [("wgrad_reduce", "wgrad_reduce"), ("conv_wgrad", "conv_wgrad"), ("conv_fwd", "conv_fwd"), ("conv_thin", "conv_fwd"),
            ("conv_narrow", "conv_fwd"), ("conv1x1_up2x", "conv_up_fwd"), ("xcorr_partial", "conv_stats_xcorr"),
            ("bn_act_bwd_apply", "bn_bwd_apply"), ("bn_act_bwd_reduce", "bn_bwd_reduce"), ("bn_fused_fwd", "bn_fwd"),
            ("bn_act_fwd", "bn_fwd"), ("maxpool_bwd", "maxpool_bwd"), ("maxpool_fwd", "maxpool_fwd"),
            ("upsample_bwd", "upsample_bwd"), ("upsample_fwd", "upsample_fwd"), ("pack_weights", "pack_weights"),
            ("adamw_step", "adamw_step")]


def _source_hash():
    try:
        from fedicra_amd import _lib
        return _lib.source_hash()
    except Exception as e:  # noqa: BLE001
        return "unknown: " + repr(e)[:80]


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


def mfma(path):
    """Per family, from ONE pass holding the four counters: per-dispatch sums over every counter instance rocpd stores.
    SQ_VALU_MFMA_BUSY_CYCLES counts cycles an MFMA is executing, per SIMD, summed over the SIMDs of the sampled shader engines;
    SQ_BUSY_CYCLES counts cycles the SQ of a sampled engine had a wave; GRBM_GUI_ACTIVE the GPU-active cycles.  Reported:
    the raw per-launch sums, the number of rows (instances) per dispatch, and
        mfma_busy_pct = 100 x MFMA_BUSY / (GRBM_GUI_ACTIVE_per_instance x 256 CUs x 4 SIMDs)
    (rocprof-compute's "MFMA utilisation"); `mops_flops_per_launch` = 512 x SQ_INSTS_VALU_MFMA_MOPS_BF16 is the bf16 matrix
    work the hardware saw, to be held against the algorithmic flops (a ratio far from 1 means the counter samples a part of
    the chip: the percentage is then scaled by the same ratio and says so)."""
    c = sqlite3.connect(path)
    q = ("select s.kernel_name, p.name, e.value, d.id from rocpd_pmc_event e join rocpd_info_pmc p on e.pmc_id = p.id "
         "join rocpd_kernel_dispatch d on d.event_id = e.event_id join rocpd_info_kernel_symbol s on d.kernel_id = s.id")
    acc = defaultdict(lambda: defaultdict(float))
    rows = defaultdict(lambda: defaultdict(int))
    disp = defaultdict(set)
    for name, pname, val, did in c.execute(q):
        f = family(name)
        if f:
            acc[f][pname] += val
            rows[f][pname] += 1
            disp[f].add(did)
    out = {}
    for f, d in acc.items():
        n = max(len(disp[f]), 1)
        inst = {k: rows[f][k] / float(n) for k in d}
        gui = d.get("GRBM_GUI_ACTIVE", 0.0) / max(rows[f].get("GRBM_GUI_ACTIVE", 1), 1)          # cycles per dispatch, one instance
        busy = d.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / n
        out[f] = {"launches": n, "mfma_busy_cycles_per_launch": busy, "gui_active_cycles_per_launch": gui,
                  "sq_busy_cycles_per_launch": d.get("SQ_BUSY_CYCLES", 0.0) / n,
                  "mops_flops_per_launch": 512.0 * d.get("SQ_INSTS_VALU_MFMA_MOPS_BF16", 0.0) / n,
                  "counter_rows_per_dispatch": inst,
                  "mfma_busy_pct": round(100.0 * busy / max(gui * 256 * 4, 1.0), 3)}
    return out


def process_totals(log_path):
    """roofline.process_totals of the counted process, from its stdout (the last line that parses as the bench's JSON)."""
    if not log_path or not os.path.exists(log_path):
        return {}
    for ln in reversed(open(log_path, errors="replace").read().splitlines()):
        ln = ln.strip()
        if ln.startswith("{") and "process_totals" in ln:
            try:
                return json.loads(ln)["roofline"]["process_totals"]
            except (ValueError, KeyError):
                continue
    return {}


def main(fetch_db, write_db, bench_args="", key="c3-unet_lc-12x3x512-bf16", mfma_db=None, bench_log=None):
    rd, wr = collect(fetch_db, "FETCH_SIZE"), collect(write_db, "WRITE_SIZE")
    totals = process_totals(bench_log)
    out = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) --kernel-trace -- python bench.py "
                     + bench_args + "; 1x MI355X (tools/pmc_round.sh)",
           "workload_key": key, "source_hash": _source_hash(),
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
        t = totals.get(fam)
        if t and t.get("algorithmic_bytes"):
            hbm_total = 2.0 * kb * 1024.0 + kbw * 1024.0
            out["families"][fam].update({
                "hbm_bytes_total": int(hbm_total), "algorithmic_bytes_total": int(t["algorithmic_bytes"]),
                "algorithmic_launches": int(t["launches"]),
                "traffic_over_algorithmic": round(hbm_total / t["algorithmic_bytes"], 4),
                "population": f"every launch of the counted process: {n} dispatches counted (read pass), {nw} (write pass), "
                              f"{int(t['launches'])} C-ABI launches priced -- totals over totals, same process"})
    if mfma_db:
        out["mfma_source"] = ("rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_BF16 "
                              "--kernel-trace (its own pass)")
        for fam, v in mfma(mfma_db).items():
            out["families"].setdefault(fam, {})["mfma"] = v
    json.dump(out, sys.stdout, indent=1)


main(*sys.argv[1:])
