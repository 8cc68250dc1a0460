#!/usr/bin/env python
"""The 3x3x3 convolutions of unet_3D below 128^3 (the general one-launch form, conv_fwd_ws_kernel with depth taps as channel groups),
forward and input gradient: us per launch, hipGraph-timed, against the MFMA / HBM roofline.  Knobs from the environment (FI_WS3D_NF,
FI_WS3D_PW, FI_WS3D_CK, FI_WS3D_WGS): one process per setting.    python tools/c3g_bench.py [--edges 64,32] [--reps 10]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fedicra_amd import _lib as L  # noqa: E402
from tools.c3s_bench import timeit  # noqa: E402

# (edge, c0, c1, co0, co1, kind, calls per iteration)
LAYERS = [(64, 16, 0, 32, 0, "fwd", 1), (64, 32, 0, 32, 0, "fwd", 2), (64, 32, 64, 32, 0, "fwd", 1),
          (64, 32, 0, 16, 0, "dgrad", 1), (64, 32, 0, 32, 0, "dgrad", 2), (64, 32, 0, 32, 64, "dgrad", 1),
          (32, 32, 0, 64, 0, "fwd", 1), (32, 64, 0, 64, 0, "fwd", 2), (32, 64, 128, 64, 0, "fwd", 1),
          (32, 64, 0, 32, 0, "dgrad", 1), (32, 64, 0, 64, 0, "dgrad", 2), (32, 64, 0, 64, 128, "dgrad", 1),
          (16, 64, 0, 128, 0, "fwd", 1), (16, 128, 0, 128, 0, "fwd", 2), (16, 128, 256, 128, 0, "fwd", 1),
          (16, 128, 0, 64, 0, "dgrad", 1), (16, 128, 0, 128, 0, "dgrad", 2), (16, 128, 0, 128, 256, "dgrad", 1),
          (8, 128, 0, 256, 0, "fwd", 1), (8, 256, 0, 256, 0, "fwd", 1), (8, 256, 0, 128, 0, "dgrad", 1), (8, 256, 0, 256, 0, "dgrad", 1)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--edges", default="")
    ap.add_argument("--reps", type=int, default=10)
    a = ap.parse_args()
    edges = {int(e) for e in a.edges.split(",") if e}
    td, dev, N = torch.bfloat16, "cuda", 2
    tot = ideal = 0.0
    for S, c0, c1, co0, co1, kind, calls in LAYERS:
        if edges and S not in edges:
            continue
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
        us = timeit(fn, a.reps)
        vox = N * S ** 3
        gf = 2.0 * vox * cin * cout * 27 / 1e9
        idl = max(gf * 1e9 / 2.5e15, vox * (cin + cout) * 2 / 8e12) * 1e6
        tot += us * calls
        ideal += idl * calls
        print(f"{kind:5s} 2 x {S:3d}^3 {cin:3d}->{cout:3d}: {us:7.1f} us  ideal {idl:5.1f}  frac {idl / us:5.2f}  {gf / us * 1e3:6.1f} TF/s  x{calls}")
    env = {k: v for k, v in os.environ.items() if k.startswith("FI_WS3D")}
    print(f"TOTAL {tot:.0f} us per iteration, ideal {ideal:.0f}, frac {ideal / tot:.3f}  ({env})")


if __name__ == "__main__":
    main()
