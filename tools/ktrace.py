#!/usr/bin/env python
"""Per-workgroup timeline of the forward/dgrad conv kernel (debug build: tools/build_variant.sh with EXTRA=-DFI_TRACE).

    EXTRA=-DFI_TRACE tools/build_variant.sh WORK trace && python tools/ktrace.py variants/trace.so

For every bench layer: launch once with a trace buffer, then report the kernel's wall span, the median duration of
each phase inside a workgroup (s_memtime ticks converted with the 100 MHz wall clock), and how many workgroups were
alive at the median point of the launch.
"""
import ctypes as C
import sys

import numpy as np
import torch

from kbench import LAYERS, FiConv, p


def main(path, kind="fwd", dtype="bf16", N=12):
    lib = C.CDLL(path)
    td = torch.bfloat16 if dtype == "bf16" else torch.float32
    di = 1 if dtype == "bf16" else 0
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    trace = torch.zeros(1 << 16, 8, dtype=torch.int64, device="cuda")
    print("layer                blocks  span_us | median per-workgroup us: stage1  k-loop(rest)  epilogue  stats  total | alive@mid  t_first_done")
    for (H, c0, c1, cout, k) in LAYERS:
        cin = c0 + c1
        x0 = torch.randn(N, H, H, c0, device="cuda").to(td)
        x1 = torch.randn(N, H, H, c1, device="cuda").to(td) if c1 else None
        w = torch.randn(cout, k * k, cin, device="cuda").to(td)
        y = torch.empty(N, H, H, cout, device="cuda", dtype=td)
        stats = torch.zeros(8 * cout * 2, dtype=torch.float64, device="cuda")
        d = FiConv(di, N, H, H, k, c0, c1, cout, 0, 0, 0, 0)
        dy = torch.randn(N, H, H, cout, device="cuda").to(td)
        dw = torch.zeros(cout, k, k, cin, device="cuda")
        db = torch.zeros(cout, device="cuda")
        lib.fi_conv2d_wgrad_workspace.restype = C.c_long
        nb = lib.fi_conv2d_wgrad_workspace(C.byref(d))
        ws = torch.empty(max(nb, 4) // 4, device="cuda")
        for rep in range(3):
            trace.zero_()
            lib.fi_debug_set_trace(C.c_void_p(trace.data_ptr() if rep == 2 else 0))
            if kind == "fwd":
                rc = lib.fi_conv2d_fwd(C.byref(d), p(x0), p(x1), p(w), None, p(y), None, p(stats), st)
            else:   # phases: first tile staged / tile loop / wave combine / partial-slice store
                rc = lib.fi_conv2d_wgrad_partial(C.byref(d), p(x0), p(x1), p(dy), 1, p(ws), C.c_long(nb), C.byref(C.c_int(0)), C.byref(C.c_long(0)), st)
            assert rc == 0, rc
            torch.cuda.synchronize()
        lib.fi_debug_set_trace(C.c_void_p(0))
        t = trace.cpu().numpy()
        t = t[t[:, 0] != 0]
        nb = len(t)
        # slots: 0 entry, 1 first chunk staged, 3 K loop done, 4 stores issued, 5 end (s_memtime); 6 / 2 = 100 MHz wall
        # clock at entry / end (comparable across CUs), 7 = HW_ID | XCC_ID << 32
        life_w = (t[:, 2] - t[:, 6]) / 100.0
        life_m = (t[:, 5] - t[:, 0]).astype(np.float64)
        tpu = float(np.median(life_m[life_w > 0] / life_w[life_w > 0]))
        ph = lambda a, b: np.median((t[:, b] - t[:, a]) / tpu)
        w0 = t[:, 6].min()
        span = (t[:, 2].max() - w0) / 100.0
        mid = w0 + (t[:, 2].max() - w0) // 2
        alive = int(((t[:, 6] <= mid) & (t[:, 2] >= mid)).sum())
        first_done = (t[:, 2].min() - w0) / 100.0
        last_start = (t[:, 6].max() - w0) / 100.0
        print(f"{H:4d} {cin:3d}->{cout:3d} k{k}   {nb:6d}  {span:7.1f} | {ph(0,1):6.2f} {ph(1,3):6.2f} {ph(3,4):6.2f} {ph(4,5):6.2f} {ph(0,5):6.2f} | "
              f"{alive:6d}  {first_done:6.2f}  last start {last_start:6.2f}  (ticks/us {tpu:.0f})")


if __name__ == "__main__":
    main(sys.argv[1], *(sys.argv[2:3]))
