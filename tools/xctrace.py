#!/usr/bin/env python
"""Per-workgroup phase times of xcorr_partial_kernel (an -DXC_TRACE build: FEDICRA_HIP_LIB=variants/xc_trace.so)."""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fedicra_amd import _lib as L  # noqa: E402

N, H, W, groups, cout = 84, 128, 128, 7, 512
g = torch.Generator().manual_seed(3)
y = (torch.randn(N, H, W, 64, generator=g) * 1.3 + 0.2).to(torch.bfloat16).cuda()
coef = torch.stack([torch.rand(groups, 64, generator=g) + 0.5, torch.randn(groups, 64, generator=g) * 0.3]).cuda()
w = (torch.randn(cout, 3, 3, 64, generator=g) * 0.05).cuda()
wp = torch.empty(cout * 9 * 64, dtype=torch.bfloat16, device="cuda")
L.pack_weights(w, wp, cout, 9, 64, 0)
stats = torch.zeros(groups * L.STATS_SLOTS * cout * 2, dtype=torch.float64, device="cuda")
tr = torch.zeros(256 * 4, dtype=torch.int64, device="cuda")
L.lib().fi_xcorr_debug_set_trace(C.c_void_p(tr.data_ptr()))
for _ in range(3):
    L.conv2d_stats_xcorr(y, L.in_xform(coef, 0.01), wp, None, stats, groups=groups, cout=cout)
torch.cuda.synchronize()
t = tr.cpu().numpy().reshape(256, 4)[:252]
d = np.diff(t, axis=1)
tot = t[:, 3] - t[:, 0]
print("ticks (median over workgroups): prologue %d, row loop %d (%.0f per row of 43), epilogue %d" % (np.median(d[:, 0]), np.median(d[:, 1]), np.median(d[:, 1]) / 43.0, np.median(d[:, 2])))
print("per-workgroup total: min %d median %d max %d; first start .. last end %d; starts spread %d" % (tot.min(), np.median(tot), tot.max(), t[:, 3].max() - t[:, 0].min(), t[:, 0].max() - t[:, 0].min()))
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5):
    L.conv2d_stats_xcorr(y, L.in_xform(coef, 0.01), wp, None, stats, groups=groups, cout=cout)
e1.record()
torch.cuda.synchronize()
print("whole call %.1f us" % (e0.elapsed_time(e1) / 5 * 1e3))
