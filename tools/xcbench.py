#!/usr/bin/env python
"""The statistics-only head launch of the LC forwards (84 x 128^2 x 64 -> 512, 7 groups): autocorrelation form
(fi_conv2d_stats_xcorr) against the direct launch, us per call; under rocprofv3 --kernel-trace --stats the per-kernel split."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fedicra_amd import _lib as L  # noqa: E402


def main():
    N, H, W, groups, cout = 84, 128, 128, 7, 512
    g = torch.Generator().manual_seed(3)
    y = (torch.randn(N, H, W, 64, generator=g) * 1.3 + 0.2).to(torch.bfloat16).cuda()
    coef = torch.stack([torch.rand(groups, 64, generator=g) + 0.5, torch.randn(groups, 64, generator=g) * 0.3]).cuda()
    w = (torch.randn(cout, 3, 3, 64, generator=g) * 0.05).cuda()
    bias = (torch.randn(cout, generator=g) * 0.1).cuda()
    wp = torch.empty(cout * 9 * 64, dtype=torch.bfloat16, device="cuda")
    L.pack_weights(w, wp, cout, 9, 64, 0)
    w16 = torch.empty_like(wp)
    L.pack_weights(w, w16, cout, 9, 64, 2)
    wp._fi_w16 = w16
    t0 = L.in_xform(coef, 0.01)
    stats = torch.zeros(groups * L.STATS_SLOTS * cout * 2, dtype=torch.float64, device="cuda")

    def timed(fn, reps=10):
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps * 1e3

    print("xcorr  %.1f us" % timed(lambda: L.conv2d_stats_xcorr(y, t0, wp, bias, stats, groups=groups, cout=cout)))
    if "--no-direct" not in sys.argv:
        print("direct %.1f us" % timed(lambda: L.conv2d_fwd_fused(y, t0, None, None, wp, bias, None, stats, ksize=3, groups=groups, cout=cout)))


main()
