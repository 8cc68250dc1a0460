#!/usr/bin/env python
"""Known-traffic launches for calibrating rocprofv3's FETCH_SIZE / WRITE_SIZE on this box: fi_scale over a 512 MB fp32
buffer (reads 512 MB, writes 512 MB; larger than the 256 MB Infinity Cache) and the bf16 cast kernel (reads 512 MB
fp32, writes 256 MB).  Run under `rocprofv3 --pmc FETCH_SIZE --kernel-trace` (and again with WRITE_SIZE)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fedicra_amd import _lib as L
n = 128 * 1024 * 1024
x = torch.ones(n, device="cuda")
y = torch.empty(n, device="cuda")
h = torch.empty(n, device="cuda", dtype=torch.bfloat16)
for _ in range(5):
    L.scale(x, y, 2.0)
    L.cast(x, h)
torch.cuda.synchronize()
