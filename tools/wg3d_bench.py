#!/usr/bin/env python
"""Every 3x3x3 filter-gradient launch of a unet_3D iteration (2 x 128^3, /root/reference/code/networks/unet_3D.py:22-60 through
fedicra_amd/networks/unet_3D.py): fi_conv3d_wgrad_fused_partial stage 1, us per launch against the HBM / MFMA roofline.  Knobs come from
the environment (FI_WGRAD_ROWS3D_ITEMS, FI_WGRAD_ROWS3D_XCD, FI_WGRAD_ROWS64_3D ...): one process per setting."""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fedicra_amd import _lib as L  # noqa: E402
from tools.kbench2 import timeit  # noqa: E402

# (edge, c0, c1, cout, calls per iteration)
LAYERS = [(128, 16, 0, 16, 2), (128, 16, 32, 16, 1),
          (64, 16, 0, 32, 1), (64, 32, 0, 32, 2), (64, 32, 64, 32, 1),
          (32, 32, 0, 64, 1), (32, 64, 0, 64, 2), (32, 64, 128, 64, 1),
          (16, 64, 0, 128, 1), (16, 128, 0, 128, 2), (16, 128, 256, 128, 1),
          (8, 128, 0, 256, 1), (8, 256, 0, 256, 1)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=2)
    ap.add_argument("--reps", type=int, default=8)
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--edges", default="", help="comma list: only these volume edges")
    a = ap.parse_args()
    td = torch.bfloat16 if a.dtype == "bf16" else torch.float16
    edges = {int(e) for e in a.edges.split(",") if e}
    tot = ideal = 0.0
    for e, c0, c1, cout, calls in LAYERS:
        if edges and e not in edges:
            continue
        x0 = torch.randn(a.batch, e, e, e, c0, device="cuda").to(td)
        x1 = torch.randn(a.batch, e, e, e, c1, device="cuda").to(td) if c1 else None
        dy = torch.randn(a.batch, e, e, e, cout, device="cuda").to(td)
        res = L.conv3d_wgrad_fused_partial(x0, x1, dy, True, ksize=3)
        if res is None:
            print(f"{e}^3 {c0 + c1}->{cout}: not covered by the one-launch form")
            continue
        us = timeit(lambda: L.conv3d_wgrad_fused_partial(x0, x1, dy, True, ksize=3), a.reps)
        vox = a.batch * e ** 3
        gf = 2.0 * vox * (c0 + c1) * cout * 27 / 1e9
        by = vox * (c0 + c1 + cout) * 2.0
        idl = max(gf * 1e9 / 2.5e15, by / 8e12) * 1e6
        tot += us * calls
        ideal += idl * calls
        print(f"{a.batch} x {e:3d}^3 {c0 + c1:3d}->{cout:3d}: {us:8.1f} us  ideal {idl:6.1f}  frac {idl / us:5.2f}  {gf / us * 1e3:6.1f} TF/s  "
              f"slices {res[1]:4d}  x{calls}")
    env = {k: v for k, v in os.environ.items() if k.startswith("FI_WGRAD")}
    print(f"TOTAL {tot:.0f} us per iteration, ideal {ideal:.0f}, frac {ideal / tot:.3f}  ({a.dtype}, {env})")


main()
