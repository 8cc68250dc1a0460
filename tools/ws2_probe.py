#!/usr/bin/env python
"""Time a few fused / plain launches of conv_fwd_ws2_kernel under every build variant given (tools/ws2_variants.sh):
    python tools/ws2_probe.py variants/libws2dbg0.so variants/libws2dbg22.so ...
Each variant runs in its own process (FEDICRA_HIP_LIB); prints us per launch and TF/s per layer."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# (name, N, H, c0, c1, cout, fused kind or None, tr)
LAYERS = [("f64_128>128", 84, 64, 128, 0, 128, "drop", 1), ("f64_256>128", 84, 64, 128, 128, 128, "xf", 1),
          ("f32_256>256", 84, 32, 256, 0, 256, "drop", 1), ("f128_64>64", 84, 128, 64, 0, 64, "drop", 2),
          ("f128_128>64", 84, 128, 64, 64, 64, "xf", 2), ("head", 84, 128, 64, 0, 512, "head", 1),
          ("p64_128>128", 12, 64, 128, 0, 128, None, 2), ("p128_64>64", 12, 128, 64, 0, 64, None, 2)]


def child():
    import torch
    sys.path.insert(0, ROOT)
    from fedicra_amd import _lib as L
    from tools.kbench2 import timeit
    td, dev = torch.bfloat16, "cuda"
    out = []
    for name, N, H, c0, c1, cout, kind, tr in LAYERS:
        G = 7 if kind else 1
        x0 = torch.randn(N, H, H, c0, device=dev).to(td)
        x1 = torch.randn(N, H, H, c1, device=dev).to(td) if c1 else None
        wf = torch.randn(cout, 3, 3, c0 + c1, device=dev) * 0.05
        w = wf.to(td)
        w16 = torch.empty(w.numel(), dtype=td, device=dev)
        L.pack_weights(wf, w16, cout, 9, c0 + c1, 2)
        w._fi_w16 = w16
        bias = torch.randn(cout, device=dev)
        y = None if kind == "head" else torch.empty(N, H, H, cout, device=dev, dtype=td)
        st = torch.zeros(G, L.STATS_SLOTS, cout, 2, dtype=torch.float64, device=dev)
        t0 = t1 = None
        if kind:
            coef = torch.rand(2, G, c0, device=dev) + 0.5
            soff = torch.zeros(1, dtype=torch.int32, device=dev)
            drop = (L.DROP_RNG_ELEM, 0.1, 1234, None, soff) if kind == "drop" else None
            t0 = L.in_xform(coef, 0.01, drop=drop, seed_group_stride=0x10001)
            if c1:
                t1 = L.in_xform(torch.rand(2, G, c1, device=dev) + 0.5, 0.0)

        def fn():
            if kind:
                L.conv2d_fwd_fused(x0, t0, x1, t1, w, bias, y, st, ksize=3, groups=G, cout=cout)
            else:
                L.conv2d_fwd(x0, x1, w, bias, y, None, st[0], ksize=3, cout=cout)
        L.conv_tuning(7, tr, 0, int(os.environ.get("WS2_WGS", "0")))
        us = timeit(fn, 8)
        L.conv_tuning(-1)
        tf = 2.0 * N * H * H * (c0 + c1) * cout * 9 / us / 1e6
        out.append(f"{name}:{us:7.1f}us {tf:6.0f}TF")
    print(" | ".join(out), flush=True)


if __name__ == "__main__":
    if os.environ.get("WS2_CHILD"):
        child()
    else:
        for lib in sys.argv[1:]:
            env = dict(os.environ, WS2_CHILD="1", FEDICRA_HIP_LIB=os.path.abspath(lib))
            r = subprocess.run([sys.executable, os.path.abspath(__file__)], env=env, capture_output=True, text=True)
            print(f"{os.path.basename(lib):24s} {r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-300:]}", flush=True)
