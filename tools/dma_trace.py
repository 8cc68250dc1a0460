#!/usr/bin/env python
"""Per-wave timeline of conv_fwd_dma_kernel (debug build with -DFI_TRACE; csrc/conv_dma.h).
    FEDICRA_HIP_LIB=variants/trace.so python tools/dma_trace.py [--H 64 --cin 128 --cout 128 --kind drop]
Prints, in shader cycles (s_memtime), medians over the workgroups of the segments of a stage, for the waves of the two sub-tiles: DMA issue, the
taps (MFMA groups with the transform slices between them), coefficient staging / epilogue, end-of-stage wait + barrier, whole stage."""
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
    ap.add_argument("--H", type=int, default=64)
    ap.add_argument("--cin", type=int, default=128)
    ap.add_argument("--c1", type=int, default=0)
    ap.add_argument("--cout", type=int, default=128)
    ap.add_argument("--N", type=int, default=84)
    ap.add_argument("--kind", default="drop")
    a = ap.parse_args()
    td, dev = torch.bfloat16, "cuda"
    G = 7 if a.kind != "none" else 1
    x0 = torch.randn(a.N, a.H, a.H, a.cin, device=dev).to(td)
    x1 = torch.randn(a.N, a.H, a.H, a.c1, device=dev).to(td) if a.c1 else None
    wf = torch.randn(a.cout, 3, 3, a.cin + a.c1, device=dev) * 0.05
    w = wf.to(td)
    w16 = torch.empty(w.numel(), dtype=td, device=dev)
    L.pack_weights(wf, w16, a.cout, 9, a.cin + a.c1, 2)
    w._fi_w16 = w16
    bias = torch.randn(a.cout, device=dev)
    y = torch.empty(a.N, a.H, a.H, a.cout, device=dev, dtype=td)
    st = torch.zeros(G, L.STATS_SLOTS, a.cout, 2, dtype=torch.float64, device=dev)
    t0 = t1 = None
    if a.kind != "none":
        soff = torch.zeros(1, dtype=torch.int32, device=dev)
        drop = (L.DROP_RNG_ELEM, 0.1, 1234, None, soff) if a.kind == "drop" else None
        t0 = L.in_xform(torch.rand(2, G, a.cin, device=dev) + 0.5, 0.01, drop=drop, seed_group_stride=0x10001)
        if a.c1:
            t1 = L.in_xform(torch.rand(2, G, a.c1, device=dev) + 0.5, 0.0)
    trace = torch.zeros(256 * 16 * 256, dtype=torch.int64, device=dev)
    lib = L.lib()
    L.conv_tuning(7, 8, 0, 0)

    def fn():
        if a.kind != "none":
            L.conv2d_fwd_fused(x0, t0, x1, t1, w, bias, y, st, ksize=3, groups=G, cout=a.cout)
        else:
            L.conv2d_fwd(x0, x1, w, bias, y, None, st[0], ksize=3, cout=a.cout)
    for rep in range(3):
        trace.zero_()
        lib.fi_debug_set_trace(C.c_void_p(trace.data_ptr() if rep == 2 else 0))
        fn()
        torch.cuda.synchronize()
    lib.fi_debug_set_trace(C.c_void_p(0))
    L.conv_tuning(-1)
    t = trace.cpu().numpy().reshape(256, 16, 256)
    names = {(1, 2): "dma_issue", (2, 4): "taps+transform", (4, 6): "coef+epilogue", (4, 7): "coef", (6, 7): "after_epilogue", (7, 1): "wait+barrier"}
    seg = {sub: {} for sub in "AB"}
    for wg in range(256):
        for wv in range(16):
            ev = t[wg, wv]
            ev = ev[ev != 0]
            if len(ev) < 8:
                continue
            tag, tm = ev & 15, ev >> 4
            d = seg["A" if wv < 8 else "B"]
            last1 = None
            for i in range(len(ev) - 1):
                k = names.get((int(tag[i]), int(tag[i + 1])))
                if k:
                    d.setdefault(k, []).append(tm[i + 1] - tm[i])
                if tag[i] == 1:
                    if last1 is not None:
                        d.setdefault("stage", []).append(tm[i] - last1)
                    last1 = tm[i]
    q = lambda v: f"median {np.median(v):8.0f}  p10 {np.percentile(v, 10):8.0f}  p90 {np.percentile(v, 90):8.0f}  (n={len(v)})"
    print(f"layer {a.N}x{a.H}^2 {a.cin + a.c1}->{a.cout} kind {a.kind}: shader cycles")
    for sub in "AB":
        for k in ("dma_issue", "taps+transform", "coef", "coef+epilogue", "wait+barrier", "stage"):
            if k in seg[sub]:
                print(f"  sub-tile {sub} {k:14s} {q(np.array(seg[sub][k]))}")


if __name__ == "__main__":
    main()
