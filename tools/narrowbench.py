#!/usr/bin/env python
"""The launches of the layers with a <= 4-channel side (unet_lc at B x 3 x 512^2, bf16) under the narrow forms and under the
general kernels (fi_narrow_tuning 1 / 0): us per launch against the HBM floor.  python tools/narrowbench.py [--batch 12]"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fedicra_amd import _lib as L  # noqa: E402
from tools.kbench2 import timeit  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=12)
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--reps", type=int, default=6)
    ap.add_argument("--ncls", type=int, default=3)
    a = ap.parse_args()
    B, S, K = a.batch, a.size, a.ncls
    dt = torch.bfloat16
    dev = "cuda"
    xin = torch.randn(B, 3, S, S, device=dev)
    x3 = torch.randn(B, S, S, 3, device=dev).to(dt)
    x16 = torch.randn(B, S, S, 16, device=dev).to(dt)
    dy16 = torch.randn(B, S, S, 16, device=dev).to(dt)
    dyk = torch.randn(B, S, S, K, device=dev).to(dt)
    w_in = torch.randn(16 * 9 * 3, device=dev).to(dt)
    w_out = torch.randn(K * 9 * 16, device=dev).to(dt)
    b16, bk = torch.randn(16, device=dev), torch.randn(K, device=dev)
    y16 = torch.empty(B, S, S, 16, dtype=dt, device=dev)
    yk = torch.empty(B, S, S, K, dtype=torch.float32, device=dev)
    nh = torch.empty(B, S, S, 3, dtype=dt, device=dev)
    stats = torch.zeros(L.STATS_SLOTS * 16 * 2, dtype=torch.float64, device=dev)
    px = B * S * S
    cases = [
        ("nchw->nhwc 3 planes", lambda: L.nchw_to_nhwc(xin, nh), px * 3 * 6),
        ("fwd 3->16 + stats", lambda: L.conv2d_fwd(x3, None, w_in, b16, y16, None, stats, ksize=3), px * (6 + 32)),
        ("dgrad %d->16" % K, lambda: L.conv2d_fwd(dyk, None, w_out, None, y16, None, None, ksize=3), px * (2 * K + 32)),
        ("fwd 16->%d fp32" % K, lambda: L.conv2d_fwd(x16, None, w_out, bk, yk, None, None, ksize=3, y_f32=True), px * (32 + 4 * K)),
        ("wgrad 3->16", lambda: L.conv2d_wgrad_partial(x3, None, dy16, True, ksize=3), px * (6 + 32)),
        ("wgrad 16->%d" % K, lambda: L.conv2d_wgrad_partial(x16, None, dyk, True, ksize=3), px * (32 + 2 * K)),
    ]
    print(f"{B} x {S}^2, bf16; us per launch (narrow / general) and the HBM floor at 8 TB/s")
    tot = [0.0, 0.0]
    for name, fn, nbytes in cases:
        us = []
        for on in (7, 0):
            L.lib().fi_narrow_tuning(on)
            us.append(timeit(fn, a.reps))
        L.lib().fi_narrow_tuning(-1)
        floor = nbytes / 8e12 * 1e6
        tot[0] += us[0]
        tot[1] += us[1]
        print(f"{name:22s} {us[0]:8.1f} / {us[1]:8.1f} us   floor {floor:6.1f}   frac {floor / us[0]:5.2f} / {floor / us[1]:5.2f}")
    print(f"sum {tot[0]:.0f} / {tot[1]:.0f} us")


main()
