#!/usr/bin/env python
"""UpBlock's conv1x1 + bilinear x2 at the four decoder levels: the two launches (fi_conv2d_fwd[_fused] + fi_upsample2x_fwd) against
the one-launch form (fi_conv1x1_up2x_fwd, csrc/upfuse.hip) over its rows-per-workgroup setting; us per launch and GB/s of the
one-launch form's algorithmic traffic (source once, result once).
    python tools/upfbench.py [--images 84] [--size 512] [--raw 1]"""
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
    ap.add_argument("--images", type=int, default=84)
    ap.add_argument("--groups", type=int, default=7)
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--reps", type=int, default=8)
    ap.add_argument("--raw", type=int, default=1, help="1: the source is a raw convolution output (BatchNorm + LeakyReLU in the loader)")
    ap.add_argument("--rows", default="0,2,3,4,6,8")
    a = ap.parse_args()
    dt = torch.bfloat16
    for f, cin in ((16, 256), (8, 128), (4, 64), (2, 32)):
        cout, h = cin // 2, a.size // f
        x = torch.randn(a.images, h, h, cin, device="cuda").to(dt)
        wp = (torch.randn(cout, cin, device="cuda") / cin ** 0.5).to(dt)
        bias = torch.randn(cout, device="cuda")
        raw = a.raw and f != 16                       # the deepest level's source is the channel-selected tensor itself
        coef = torch.stack([1 + 0.1 * torch.randn(a.groups, cin), 0.1 * torch.randn(a.groups, cin)]).cuda() if raw else None
        y = torch.empty(a.images, h, h, cout, device="cuda", dtype=dt)
        u = torch.empty(a.images, 2 * h, 2 * h, cout, device="cuda", dtype=dt)
        t0 = None if coef is None else L.in_xform(coef, 0.01)

        def conv():
            if t0 is None:
                L.conv2d_fwd(x, None, wp, bias, y, None, None, ksize=1)
            else:
                L.conv2d_fwd_fused(x, t0, None, None, wp, bias, y, None, ksize=1, groups=a.groups, cout=cout)

        us_c = timeit(conv, a.reps)
        us_u = timeit(lambda: L.upsample2x_fwd(y, u), a.reps)
        nb = (x.numel() + u.numel()) * 2
        line = f"{a.images} x {h:3d}^2 {cin:3d}->{cout:3d} {'raw' if raw else 'act'}: conv {us_c:7.1f} + up {us_u:7.1f} = {us_c + us_u:7.1f} us | fused"
        for r in [int(v) for v in a.rows.split(",")]:
            L.upfuse_tuning(r)
            ok = L.conv1x1_up2x_fwd(x, t0, wp, bias, u, groups=a.groups)
            if not ok:
                line += f"  R{r}: n/a"
                continue
            us = timeit(lambda: L.conv1x1_up2x_fwd(x, t0, wp, bias, u, groups=a.groups), a.reps)
            line += f"  R{r}: {us:6.1f} ({nb / us / 1e3:5.0f} GB/s)"
        L.upfuse_tuning(0)
        print(line, flush=True)


main()
