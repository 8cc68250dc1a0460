#!/usr/bin/env python
"""Per-wave timeline of conv_wgrad_rows3d_kernel (debug build with -DFI_TRACE; csrc/wgrad_rows.h).
    FEDICRA_HIP_LIB=variants/trace.so python tools/rows3d_trace.py [--c0 16 --c1 32]
Prints, in s_memtime ticks, medians over the workgroups and rows 16..31 of an item of the segments of a row step, per wave: LDS stores
(with the wait for the row's loads), load issue, the MFMA loop, the barrier, the whole step."""
import argparse
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fedicra_amd import _lib as L  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--c0", type=int, default=16)
    ap.add_argument("--c1", type=int, default=32)
    ap.add_argument("--e", type=int, default=128)
    a = ap.parse_args()
    td, dev = torch.bfloat16, "cuda"
    x0 = torch.randn(2, a.e, a.e, a.e, a.c0, device=dev).to(td)
    x1 = torch.randn(2, a.e, a.e, a.e, a.c1, device=dev).to(td) if a.c1 else None
    dy = torch.randn(2, a.e, a.e, a.e, 16, device=dev).to(td)
    nwg = 1024
    trace = torch.zeros(nwg * 4 * 16 * 8, dtype=torch.int64, device=dev)
    lib = L.lib()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    for rep in range(3):
        trace.zero_()
        lib.fi_debug_set_trace(C.c_void_p(trace.data_ptr() if rep == 2 else 0))
        ev[0].record()
        res = L.conv3d_wgrad_fused_partial(x0, x1, dy, True, ksize=3)
        ev[1].record()
        torch.cuda.synchronize()
    lib.fi_debug_set_trace(C.c_void_p(0))
    print(f"{a.e}^3 {a.c0 + a.c1}->16: {ev[0].elapsed_time(ev[1]) * 1e3:.1f} us with the trace, slices {res[1]}")
    t = trace.cpu().numpy().reshape(nwg, 4, 16, 8)[:res[1]]
    names = ["stores(+vmcnt)", "load issue", "mfma loop", "barrier", "step"]
    for w in range(4):
        tw = t[:, w]
        ok = (tw[:, :, :5] > 0).all(axis=(1, 2))
        tw = tw[ok]
        seg = [tw[:, :, 1] - tw[:, :, 0], tw[:, :, 2] - tw[:, :, 1], tw[:, :, 3] - tw[:, :, 2], tw[:, :, 4] - tw[:, :, 3]]
        step = tw[:, 1:, 0] - tw[:, :-1, 0]
        line = "  ".join(f"{n} {np.median(s):7.0f} (p90 {np.percentile(s, 90):7.0f})" for n, s in zip(names, seg + [step]))
        print(f"wave {w} [{ok.sum()} wgs]: {line}")
    span = t[:, :, :, :5]
    span = span[span > 0]
    print(f"ticks spanned by the traced rows over all workgroups: {span.max() - span.min()}")


main()
