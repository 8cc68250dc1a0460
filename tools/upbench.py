#!/usr/bin/env python
"""Up-sampling kernel timing at the batched LC-forward shapes: python tools/upbench.py  (FI_UP_ROWS_MIN=99999999999 = flat form)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fedicra_amd import _lib as L
from kbench2 import timeit
for (N, h, C) in [(84, 256, 16), (84, 128, 32), (84, 64, 64), (84, 32, 128), (12, 256, 16)]:
    x = torch.randn(N, h, h, C, device="cuda").to(torch.bfloat16)
    y = torch.empty(N, 2 * h, 2 * h, C, device="cuda", dtype=torch.bfloat16)
    t = timeit(lambda: L.upsample2x_fwd(x, y), 8)
    print(f"{N:3d} x {h:3d}^2 x {C:3d}: {t:7.1f} us  {(x.numel() + y.numel()) * 2 / t / 1e3:7.1f} GB/s")
