#!/usr/bin/env python
"""The thin 128^3 layers of unet_3D alone (csrc/conv3d_stream.hip against the general one-launch form): us per launch, hipGraph-timed.
    python tools/c3s_bench.py [--size 128] [--reps 10]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fedicra_amd import _lib as L  # noqa: E402


def timeit(fn, reps):
    fn()
    torch.cuda.synchronize()
    s = torch.cuda.Stream()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(s):
        with torch.cuda.graph(g, stream=s):
            for _ in range(reps):
                fn()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=128)
    ap.add_argument("--reps", type=int, default=10)
    a = ap.parse_args()
    td, dev, S, N = torch.bfloat16, "cuda", a.size, 2
    for name, c0, c1, co0, co1, kind in (("fwd 16->16", 16, 0, 16, 0, "fwd"), ("fwd 16+32->16", 16, 32, 16, 0, "fwd"),
                                          ("dgrad 16->16", 16, 0, 16, 0, "dgrad"), ("dgrad 16->16+32", 16, 0, 16, 32, "dgrad")):
        cin, cout = c0 + c1, co0 + co1
        x0 = torch.randn(N, S, S, S, c0, device=dev).to(td)
        x1 = torch.randn(N, S, S, S, c1, device=dev).to(td) if c1 else None
        w_all = (torch.randn(cout, 9, 3, cin, device=dev) * 0.05).to(td)
        bias = torch.randn(cout, device=dev)
        st = torch.zeros(N, L.STATS_SLOTS, cout, 2, dtype=torch.float64, device=dev)
        y0 = torch.empty(N, S, S, S, co0, dtype=td, device=dev)
        y1 = torch.empty(N, S, S, S, co1, dtype=td, device=dev) if co1 else None

        def fn():
            if kind == "fwd":
                L.conv3d_fwd_fused(x0, x1, w_all, bias, y0, st, ksize=3)
            else:
                L.conv3d_dgrad_fused(x0, w_all, y0, y1, ksize=3)
        res = {}
        for on in (0, 1):
            L.conv3d_tuning(on)
            res[on] = timeit(fn, a.reps)
        L.conv3d_tuning(-1)
        nbytes = N * S ** 3 * (cin + cout) * 2
        print(f"{name:18s} general {res[0]:7.1f} us   streaming {res[1]:7.1f} us   ({res[0] / res[1]:4.2f}x, {nbytes / res[1] / 1e3:6.0f} GB/s, "
              f"roofline {nbytes / 8e12 * 1e6 / res[1]:5.3f})")


if __name__ == "__main__":
    main()
