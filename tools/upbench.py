#!/usr/bin/env python
"""Bilinear x2 up-sampling (fi_upsample2x_fwd) at the four decoder levels of the batched LC forwards: us per launch and
GB/s of the algorithmic traffic (read the low-resolution tensor once, write the result once).
    python tools/upbench.py [--images 84] [--size 512]
    python tools/upbench.py --bwd --images 12        fi_upsample2x_bwd at the four levels of a 12-image backward pass (run once
                                                     per FI_UPBWD_ROWS=0 / 1: the switch is read once per process)"""
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
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--reps", type=int, default=8)
    ap.add_argument("--bwd", action="store_true")
    ap.add_argument("--up3d", action="store_true", help="trilinear x2 forward and backward at unet_3D's four levels (2 x 128^3 input)")
    a = ap.parse_args()
    if a.up3d:
        tf = tb = 0.0
        for d, c in ((8, 256), (16, 128), (32, 64), (64, 32)):
            x = torch.randn(2, d, d, d, c, device="cuda").to(torch.bfloat16)
            y = torch.empty(2, 2 * d, 2 * d, 2 * d, c, device="cuda", dtype=torch.bfloat16)
            g = torch.randn_like(y)
            dx = torch.empty_like(x)
            uf = timeit(lambda: L.upsample3d2x_fwd(x, y), a.reps)
            ub = timeit(lambda: L.upsample3d2x_bwd(g, dx), a.reps)
            tf += uf
            tb += ub
            nb = (x.numel() + y.numel()) * 2
            print(f"up3d 2 x {d:3d}^3 x {c:3d}: fwd {uf:8.1f} us {nb / uf / 1e3:8.1f} GB/s   bwd {ub:8.1f} us {nb / ub / 1e3:8.1f} GB/s")
        print(f"up3d total fwd {tf:.1f} us  bwd {tb:.1f} us  (FI_UP3D_ROWS={os.environ.get('FI_UP3D_ROWS', '1')})")
        return
    if a.bwd:
        tot = 0.0
        for f, c in ((16, 128), (8, 64), (4, 32), (2, 16)):
            h = a.size // f
            g = torch.randn(a.images, 2 * h, 2 * h, c, device="cuda").to(torch.bfloat16)
            dx = torch.empty(a.images, h, h, c, device="cuda", dtype=torch.bfloat16)
            us = timeit(lambda: L.upsample2x_bwd(g, dx), a.reps)
            tot += us
            nb = (g.numel() + dx.numel()) * 2
            print(f"bwd {a.images} x {2 * h:3d}^2 x {c:3d} -> {h:3d}^2: {us:8.1f} us  {nb / us / 1e3:8.1f} GB/s")
        print(f"bwd total {tot:.1f} us  (FI_UPBWD_ROWS={os.environ.get('FI_UPBWD_ROWS', '1')}, FI_UPBWD_WGS={os.environ.get('FI_UPBWD_WGS', '768')})")
        return
    for f, c in ((16, 128), (8, 64), (4, 32), (2, 16)):
        h = a.size // f
        x = torch.randn(a.images, h, h, c, device="cuda").to(torch.bfloat16)
        y = torch.empty(a.images, 2 * h, 2 * h, c, device="cuda", dtype=torch.bfloat16)
        us = timeit(lambda: L.upsample2x_fwd(x, y), a.reps)
        nb = (x.numel() + y.numel()) * 2
        print(f"{a.images} x {h:3d}^2 x {c:3d} -> {2 * h:3d}^2: {us:8.1f} us  {nb / us / 1e3:8.1f} GB/s")


main()
