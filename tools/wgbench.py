#!/usr/bin/env python
"""Filter-gradient launches (fi_conv2d_wgrad_partial: stage 1, the partial slices) of the U-Net's channel-rich 3x3 layers at
12 images: us per launch and TFLOP/s.  FI_WGRAD_TR=0 python tools/wgbench.py = the 16 x 16-quadrant kernel."""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fedicra_amd import _lib as L  # noqa: E402
from tools.kbench2 import timeit  # noqa: E402

LAYERS = [(256, 32, 0, 32), (256, 32, 32, 32), (128, 32, 0, 64), (128, 64, 0, 64), (128, 64, 64, 64), (64, 64, 0, 128),
          (64, 128, 0, 128), (64, 128, 128, 128), (32, 128, 0, 256), (32, 256, 0, 256)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=12)
    ap.add_argument("--scale", type=int, default=1, help="multiply the map sizes (1 = a 512^2 input)")
    ap.add_argument("--reps", type=int, default=8)
    a = ap.parse_args()
    tot = 0.0
    for h, c0, c1, cout in LAYERS:
        h *= a.scale
        x0 = torch.randn(a.batch, h, h, c0, device="cuda").to(torch.bfloat16)
        x1 = torch.randn(a.batch, h, h, c1, device="cuda").to(torch.bfloat16) if c1 else None
        dy = torch.randn(a.batch, h, h, cout, device="cuda").to(torch.bfloat16)
        us = timeit(lambda: L.conv2d_wgrad_partial(x0, x1, dy, True, ksize=3), a.reps)
        gf = 2.0 * a.batch * h * h * (c0 + c1) * cout * 9 / 1e9
        tot += us
        print(f"{a.batch} x {h:3d}^2 {c0 + c1:3d}->{cout:3d}: {us:8.1f} us  {gf / us / 1e3:7.1f} TF/s")
    print(f"TOTAL {tot:.0f} us")


main()
