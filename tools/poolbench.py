#!/usr/bin/env python
"""PCS pooling (fi_global_avgmax[_split]) timing at the shapes unet_lc pools: python tools/poolbench.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fedicra_amd import _lib as L
from kbench2 import timeit
for (N, h, C) in [(84, 32, 256), (12, 32, 256), (84, 512, 16), (12, 512, 16)]:
    x = torch.randn(N, h, h, C, device="cuda").to(torch.bfloat16)
    avg, mx = torch.empty(N, C, device="cuda"), torch.empty(N, C, device="cuda")
    am = torch.empty(N, C, dtype=torch.int32, device="cuda")
    t = timeit(lambda: L.global_avgmax(x, avg, mx, am), 8)
    t0 = timeit(lambda: L.lib().fi_global_avgmax(L.dt(x.dtype), L.ptr(x), L.ptr(avg), L.ptr(mx), L.ptr(am), N, h * h, C, L.stream()), 8)
    print(f"{N:3d} x {h:3d}^2 x {C:3d}: split {t:7.1f} us  one-workgroup {t0:7.1f} us  {x.numel() * 2 / t / 1e3:7.1f} GB/s")
