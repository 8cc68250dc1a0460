#!/usr/bin/env python
"""Per-layer forward-kernel sweep at the bench shapes (BASELINE configs[2]: unet_lc, 12 x C x 512 x 512): the one-tile kernel
against the persistent kernel's slab widths / chunks / residencies, switched in-process through fi_conv_tuning.

    python tools/kbench2.py [--size 512] [--reps 8] [--groups 7] [--only plain|fused]

For every conv shape of a forward: microseconds per launch (hipGraph-timed), as a plain launch over 12 images (the
iteration's own forward / dgrad geometry) and as a fused launch over groups x 12 images (the batched LC forwards)."""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fedicra_amd import _lib as L  # noqa: E402

# (H/size, c0, c1, cout, kind): 3x3 convs of unet_lc at full size = 1.0; kind = loader of the fused form
LAYERS = [(1, 16, 0, 16, "drop"), (0.5, 16, 0, 32, "pool"), (0.5, 32, 0, 32, "drop"), (0.25, 32, 0, 64, "pool"),
          (0.25, 64, 0, 64, "drop"), (0.125, 64, 0, 128, "pool"), (0.125, 128, 0, 128, "drop"),
          (0.0625, 128, 0, 256, "pool"), (0.0625, 256, 0, 256, "drop"),
          (0.125, 128, 128, 128, "xf"), (0.125, 128, 0, 128, "xf"), (0.25, 64, 64, 64, "xf"), (0.25, 64, 0, 64, "xf"),
          (0.5, 32, 32, 32, "xf"), (0.5, 32, 0, 32, "xf"), (1, 16, 16, 16, "xf"), (1, 16, 0, 16, "xf"),
          (0.25, 64, 0, 512, "head"),
          (0.125, 64, 0, 128, "xf"), (0.0625, 128, 0, 256, "xf")]   # (the pooled DownBlock inputs as the batched launches see them)


def timeit(fn, reps):
    fn()
    torch.cuda.synchronize()
    s = torch.cuda.Stream()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(s):
        with torch.cuda.graph(g, stream=s):
            for _ in range(reps):
                fn()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--reps", type=int, default=8)
    ap.add_argument("--groups", type=int, default=7)
    ap.add_argument("--batch", type=int, default=12)
    ap.add_argument("--only", default="")
    ap.add_argument("--layers", default="", help="comma-separated layer indices (LAYERS order) to restrict the sweep to")
    ap.add_argument("--ws", action="store_true", help="only the one-tile kernel vs the wave-specialised kernel")
    ap.add_argument("--thin", action="store_true", help="only the one-tile kernel vs the thin-layer kernel, thin layers only")
    ap.add_argument("--ws2", action="store_true", help="one-tile kernel vs the library's old choice (FI_V2 rule) vs the 64x64-wave-tile kernel")
    ap.add_argument("--cfgs", default="", help="comma-separated config names to keep (v1 always runs)")
    a = ap.parse_args()
    td = torch.bfloat16
    dev = "cuda"
    configs = [("v1", (0, 0, 0, 0))] + [(f"nf{nf}ck{ck}w{w}", (1, nf, ck, w)) for nf in (1, 2, 4) for ck in (16, 32)
                                        for w in (2, 4)] + [(f"thin_w{w}", (3, 0, 0, w)) for w in (2, 4, 8)] + \
        [(f"ws{pw}_nf{nf}ck{ck}", ({4: 4, 8: 5, 44: 6}[pw], nf, ck, 1)) for pw in (8, 44) for nf in (2, 4) for ck in (16, 32)]
    if a.thin:
        configs = [c for c in configs if c[0] == "v1" or c[0].startswith("thin")]
    if a.ws:
        configs = [c for c in configs if c[0] == "v1" or c[0].startswith("ws")]
    if a.ws2:
        a.ws = True
        configs = [("v1", (0, 0, 0, 0)), ("lib", (-1, 0, 0, 0)), ("old", (2, 0, 0, 0)), ("ws2_16", (7, 1, 0, 0)), ("ws2_32", (7, 2, 0, 0)),
                   ("ws2_16w2", (7, 1, 0, 2)), ("ws2_32w2", (7, 2, 0, 2)), ("ws2_res", (7, 4, 0, 0)), ("ws2_resw2", (7, 4, 0, 2)),
                   ("dma", (7, 8, 0, 0))]
    if a.cfgs:
        configs = [c for c in configs if c[0] == "v1" or c[0] in a.cfgs.split(",")]
    tot = {}
    for mode in ("plain", "fused"):
        if a.only and a.only != mode:
            continue
        print(f"== {mode}: us per launch; best persistent config vs the one-tile kernel")
        for li, (f, c0, c1, cout, kind) in enumerate(LAYERS):
            if a.layers and str(li) not in a.layers.split(","):
                continue
            if a.thin and (c0 + c1 > 32 or cout > 32):
                continue
            if a.ws and (c0 + c1 < 32 or cout <= 16):
                continue
            H = int(a.size * f)
            G = a.groups if mode == "fused" else 1
            N = a.batch * G
            pool = kind == "pool" and mode == "fused"
            hs = 2 * H if pool else H
            x0 = torch.randn(N, hs, hs, c0, device=dev).to(td)
            x1 = torch.randn(N, H, H, c1, device=dev).to(td) if c1 else None
            wf = torch.randn(cout, 3, 3, c0 + c1, device=dev) * 0.05
            w = wf.to(td)
            if a.ws2 and L.conv_weight_chunk16(td, 3, c0 + c1, cout):
                w16 = torch.empty(w.numel(), dtype=td, device=dev)
                L.pack_weights(wf, w16, cout, 9, c0 + c1, 2)
                w._fi_w16 = w16
            bias = torch.randn(cout, device=dev)
            y = None if kind == "head" else torch.empty(N, H, H, cout, device=dev, dtype=td)
            st = torch.zeros(G, L.STATS_SLOTS, cout, 2, dtype=torch.float64, device=dev)
            t0 = t1 = None
            if mode == "fused":
                coef = torch.rand(2, G, c0, device=dev) + 0.5
                soff = torch.zeros(1, dtype=torch.int32, device=dev)
                drop = (L.DROP_RNG_ELEM, 0.1, 1234, None, soff) if kind == "drop" else None
                t0 = L.in_xform(coef, 0.01, pool=pool, drop=drop, seed_group_stride=0x10001)

            def fn():
                if mode == "fused":
                    L.conv2d_fwd_fused(x0, t0, x1, t1, w, bias, y, st, ksize=3, groups=G, cout=cout)
                else:
                    L.conv2d_fwd(x0, x1, w, bias, y, None, st[0], ksize=3, cout=cout)
            res = {}
            for name, cfg in configs:
                if cfg[0] == 1 and cfg[1] > 1 and (cfg[1] // 2) * 16 >= cout:
                    continue
                if cfg[0] == 1 and pool and cfg[2] == 32:
                    continue
                if cfg[0] == 3 and (c0 + c1 > 32 or cout > 32):
                    continue
                if cfg[0] == 7 and (not hasattr(w, "_fi_w16") or pool or (cfg[1] in (1, 8) and cout % 128) or (cfg[1] == 2 and cout % 64)):
                    continue
                if cfg[0] == 7 and cfg[1] == 4 and not (cout in (32, 64) and 78336 + (c0 + c1) * cout * 18 + 768 <= 160 * 1024):
                    continue
                if cfg[0] in (4, 5, 6) and (c0 + c1 < 32 or cout <= 16 or (cfg[1] == 4 and cout <= 32) or (pool and cfg[2] == 32)):
                    continue
                L.conv_tuning(*cfg)
                try:
                    res[name] = timeit(fn, a.reps)
                except L.FiError:
                    pass
            L.conv_tuning(-1)
            best = min((v, k) for k, v in res.items() if k != "v1")
            gflop = 2.0 * N * H * H * (c0 + c1) * cout * 9 / 1e9
            nbytes = (x0.numel() + (0 if x1 is None else x1.numel()) + (0 if y is None else y.numel())) * 2
            key = f"{H:4d} {c0 + c1:3d}->{cout:3d} {kind:5s}"
            tot.setdefault(mode, [0.0, 0.0])
            tot[mode][0] += res["v1"]
            tot[mode][1] += min(best[0], res["v1"])
            print(f"{key}  v1 {res['v1']:8.1f}  best {best[1]:10s} {best[0]:8.1f}  ({res['v1'] / best[0]:4.2f}x)  "
                  f"{gflop / best[0] / 1e3:7.1f} TF/s {nbytes / best[0] / 1e3:7.1f} GB/s | "
                  + " ".join(f"{k}:{v:.0f}" for k, v in res.items() if k != "v1"))
        print(f"TOTAL {mode}: v1 {tot[mode][0]:.0f} us -> best-per-layer {tot[mode][1]:.0f} us")


if __name__ == "__main__":
    main()
