#!/usr/bin/env python
"""One conv layer, a few launches per forward-kernel configuration: the target of rocprofv3 --pmc runs (tools/kprobe.sh).
    python tools/kprobe.py --H 128 --cin 64 --cout 512 [--groups 1] [--cfg v1 nf4ck32w2 ...] [--stats-only]"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fedicra_amd import _lib as L  # noqa: E402


def parse_cfg(s):
    if s == "v1":
        return (0, 0, 0, 0)
    if s == "auto":                                     # the library's own per-layer choice
        return (-1, 0, 0, 0)
    if s == "thin":
        return (3, 0, 0, 0)
    import re
    if s == "ws2r":                                      # ... with the whole filter resident in LDS (Cout 32 / 64)
        return (7, 4, 0, 0)
    m = re.fullmatch(r"ws2t(\d+)", s)                    # 64x64-wave-tile kernel with 16- / 32-row tiles
    if m:
        return (7, {16: 1, 32: 2}[int(m.group(1))], 0, 0)
    m = re.fullmatch(r"ws(\d+)nf(\d)", s)               # wave-specialised: ws8nf4 / ws44nf4 ...
    if m:
        return ({4: 4, 8: 5, 44: 6}[int(m.group(1))], int(m.group(2)), 0, 0)
    m = re.fullmatch(r"nf(\d)ck(\d+)w(\d)", s)
    return (1, int(m.group(1)), int(m.group(2)), int(m.group(3)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--H", type=int, default=128)
    ap.add_argument("--cin", type=int, default=64)
    ap.add_argument("--c1", type=int, default=0)
    ap.add_argument("--cout", type=int, default=512)
    ap.add_argument("--batch", type=int, default=12)
    ap.add_argument("--groups", type=int, default=1)
    ap.add_argument("--fused", action="store_true")
    ap.add_argument("--stats-only", action="store_true")
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--cfg", nargs="+", default=["v1", "nf4ck32w2"])
    a = ap.parse_args()
    td, dev = torch.bfloat16, "cuda"
    G, N, H = a.groups, a.batch * a.groups, a.H
    x0 = torch.randn(N, H, H, a.cin, device=dev).to(td)
    x1 = torch.randn(N, H, H, a.c1, device=dev).to(td) if a.c1 else None
    wf = torch.randn(a.cout, 3, 3, a.cin + a.c1, device=dev) * 0.05
    w = wf.to(td)
    if L.conv_weight_chunk16(td, 3, a.cin + a.c1, a.cout):
        w16 = torch.empty(w.numel(), dtype=td, device=dev)
        L.pack_weights(wf, w16, a.cout, 9, a.cin + a.c1, 2)
        w._fi_w16 = w16
    bias = torch.randn(a.cout, device=dev)
    y = None if a.stats_only else torch.empty(N, H, H, a.cout, device=dev, dtype=td)
    st = torch.zeros(G, L.STATS_SLOTS, a.cout, 2, dtype=torch.float64, device=dev)
    t0 = L.in_xform(torch.rand(2, G, a.cin, device=dev) + 0.5, 0.01) if a.fused else None
    for c in a.cfg:
        L.conv_tuning(*parse_cfg(c))
        for _ in range(a.reps):
            if a.fused:
                L.conv2d_fwd_fused(x0, t0, x1, None, w, bias, y, st, ksize=3, groups=G, cout=a.cout)
            else:
                L.conv2d_fwd(x0, x1, w, bias, y, None, st[0], ksize=3, cout=a.cout)
        torch.cuda.synchronize()
    L.conv_tuning(-1)


main()
