#!/usr/bin/env python
"""Every filter-gradient launch of the FedICRA body-phase iteration / the ALA batch (unet_lc at 12 x 3 x 512^2, bf16):
fi_conv2d_wgrad_partial stage 1, us per launch against the HBM / MFMA roofline.  Knobs come from the environment
(FI_WGRAD_BLOCKS, FI_WGRAD_BLOCKS_THIN): one process per setting."""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fedicra_amd import _lib as L  # noqa: E402
from tools.kbench2 import timeit  # noqa: E402

# (map size, c0, c1, cout, ksize, calls per body-phase iteration)
LAYERS = [(512, 3, 0, 16, 3, 1), (512, 16, 0, 16, 3, 2), (512, 16, 16, 16, 3, 1), (512, 16, 0, 3, 3, 1),
          (256, 16, 0, 32, 3, 1), (256, 32, 0, 32, 3, 2), (256, 32, 32, 32, 3, 1), (256, 32, 0, 16, 1, 1),
          (128, 32, 0, 64, 3, 1), (128, 64, 0, 64, 3, 2), (128, 64, 64, 64, 3, 1), (128, 64, 0, 32, 1, 1),
          (64, 64, 0, 128, 3, 1), (64, 128, 0, 128, 3, 2), (64, 128, 128, 128, 3, 1), (64, 128, 0, 64, 1, 1),
          (32, 128, 0, 256, 3, 1), (32, 256, 0, 256, 3, 1), (32, 256, 0, 128, 1, 1)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=12)
    ap.add_argument("--reps", type=int, default=6)
    ap.add_argument("--rows", type=int, default=-1, help="fi_wgrad_tuning bit mask: 0 = tile kernels only, 1 = row-streaming kernel on the thin layers, "
                    "2 = its 64 x 64-channel-tile form on the channel-rich layers, -1 = the environment defaults")
    ap.add_argument("--full", type=int, default=0, help="1: time fi_conv2d_wgrad (stage 1 + the single-tensor reduce) instead of stage 1 alone")
    ap.add_argument("--min-c", type=int, default=0, help="only layers with at least this many channels on both sides")
    a = ap.parse_args()
    L.lib().fi_wgrad_tuning(a.rows)
    tot = ideal = 0.0
    for h, c0, c1, cout, ks, calls in LAYERS:
        if min(c0 + c1, cout) < a.min_c:
            continue
        x0 = torch.randn(a.batch, h, h, c0, device="cuda").to(torch.bfloat16)
        x1 = torch.randn(a.batch, h, h, c1, device="cuda").to(torch.bfloat16) if c1 else None
        dy = torch.randn(a.batch, h, h, cout, device="cuda").to(torch.bfloat16)
        if a.full:
            dw = torch.zeros(cout, ks, ks, c0 + c1, device="cuda")
            db = torch.zeros(cout, device="cuda")
            us = timeit(lambda: L.conv2d_wgrad(x0, x1, dy, dw, db, ksize=ks), a.reps)
        else:
            us = timeit(lambda: L.conv2d_wgrad_partial(x0, x1, dy, True, ksize=ks), a.reps)
        gf = 2.0 * a.batch * h * h * (c0 + c1) * cout * ks * ks / 1e9
        by = a.batch * h * h * (c0 + c1 + cout) * 2.0
        idl = max(gf * 1e9 / 2.5e15, by / 8e12) * 1e6
        tot += us * calls
        ideal += idl * calls
        print(f"{a.batch} x {h:3d}^2 {c0 + c1:3d}->{cout:3d} k{ks}: {us:8.1f} us  ideal {idl:6.1f}  frac {idl / us:5.2f}  x{calls}")
    print(f"TOTAL {tot:.0f} us per iteration, ideal {ideal:.0f}, frac {ideal / tot:.3f}  "
          f"(FI_WGRAD_BLOCKS={os.environ.get('FI_WGRAD_BLOCKS', '512')} THIN={os.environ.get('FI_WGRAD_BLOCKS_THIN', '-')} rows={a.rows} full={a.full} "
          f"ROWS64_WGS={os.environ.get('FI_WGRAD_ROWS64_WGS', '-')} TILE={os.environ.get('FI_WGRAD_ROWS64_TILE', '-')})")


if __name__ == "__main__":
    main()
