#!/usr/bin/env python
"""Per-wave stage timeline of conv_fwd_ws_kernel on a 3D layer (debug build with -DFI_TRACE; csrc/conv_impl.h FI_TWS).
    FEDICRA_HIP_LIB=variants/trace.so python tools/ws_trace.py [--e 64 --cin 32 --cout 32]
Medians over workgroups and stages 4 .. 19 of a run, in s_memtime ticks: producers (issue | commit incl. its wait for the loads | barrier),
consumers (MFMAs | epilogue when the item ends | barrier), whole stage."""
import argparse
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fedicra_amd import _lib as L  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--e", type=int, default=64)
    ap.add_argument("--cin", type=int, default=32)
    ap.add_argument("--cout", type=int, default=32)
    a = ap.parse_args()
    td, dev, N, S = torch.bfloat16, "cuda", 2, a.e
    x0 = torch.randn(N, S, S, S, a.cin, device=dev).to(td)
    w_all = (torch.randn(a.cout, 9, 3, a.cin, device=dev) * 0.05).to(td)
    bias = torch.randn(a.cout, device=dev)
    st = torch.zeros(N, L.STATS_SLOTS, a.cout, 2, dtype=torch.float64, device=dev)
    y0 = torch.empty(N, S, S, S, a.cout, dtype=td, device=dev)
    nwg = 256
    trace = torch.zeros(nwg * 12 * 16 * 8, dtype=torch.int64, device=dev)
    lib = L.lib()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    for rep in range(3):
        trace.zero_()
        lib.fi_debug_set_trace(C.c_void_p(trace.data_ptr() if rep == 2 else 0))
        ev[0].record()
        L.conv3d_fwd_fused(x0, None, w_all, bias, y0, st, ksize=3)
        ev[1].record()
        torch.cuda.synchronize()
    lib.fi_debug_set_trace(C.c_void_p(0))
    print(f"{S}^3 {a.cin}->{a.cout}: {ev[0].elapsed_time(ev[1]) * 1e3:.1f} us with the trace")
    t = trace.cpu().numpy().reshape(nwg, 12, 16, 8)
    for w in range(12):
        tw = t[:, w]
        ok = (tw[:, :, :4] > 0).all(axis=(1, 2))
        if not ok.any():
            continue
        tw = tw[ok]
        seg = [tw[:, :, 1] - tw[:, :, 0], tw[:, :, 2] - tw[:, :, 1], tw[:, :, 3] - tw[:, :, 2]]
        step = tw[:, 1:, 0] - tw[:, :-1, 0]
        role = "consumer" if w < 4 else "producer"
        names = ["mfma", "epilogue", "barrier"] if w < 4 else ["issue", "commit", "barrier"]
        line = "  ".join(f"{n} {np.median(s):6.0f} (p90 {np.percentile(s, 90):6.0f})" for n, s in zip(names + ["stage"], seg + [step]))
        print(f"wave {w:2d} {role} [{ok.sum()} wgs]: {line}")


main()
