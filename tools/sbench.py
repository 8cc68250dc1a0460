#!/usr/bin/env python
"""Micro-benchmark of the streaming (HBM-bound) kernels at the U-Net(1,2) 12x1x256x256 shapes: fused BN forward,
BN backward reduce / apply, max-pool and bilinear up-sampling forward / backward.

    python tools/sbench.py [libA.so libB.so@ENV=VAL ...]        (default: the in-tree library)

Prints microseconds per launch and the algorithmic GB/s (bytes the op must move / time) per level.
"""
import argparse
import ast
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

LEVELS = [(256, 16), (128, 32), (64, 64), (32, 128), (16, 256)]      # (H = W, channels) of the five U-Net levels


def child(dtype, reps, N=12):
    import torch
    from fedicra_amd import _lib as L
    td = torch.bfloat16 if dtype == "bf16" else torch.float32
    esz = 2 if dtype == "bf16" else 4

    def timeit(fn):
        # the launches are captured into one hipGraph and replayed: per-launch host overhead (ctypes, Python) would
        # otherwise hide anything below ~8 us
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

    out = {}
    for H, Cc in LEVELS:
        y = torch.randn(N, H, H, Cc, device="cuda").to(td)
        dz = torch.randn(N, H, H, Cc, device="cuda").to(td)
        z = torch.empty_like(y)
        dy = torch.empty_like(y)
        f = lambda *s: torch.randn(*s, device="cuda")
        gamma, beta, rmean, rvar = f(Cc), f(Cc), torch.zeros(Cc, device="cuda"), torch.ones(Cc, device="cuda")
        nbt = torch.zeros(1, dtype=torch.int64, device="cuda")
        coef = torch.zeros(4, Cc, device="cuda")
        stats = torch.zeros(L.STATS_SLOTS * Cc * 2, dtype=torch.float64, device="cuda")
        stats[:Cc * 2:2] = 0.1 * N * H * H
        stats[1:Cc * 2:2] = 1.0 * N * H * H
        sums = torch.zeros(L.STATS_SLOTS * Cc * 2, dtype=torch.float64, device="cuda")
        dgam, dbet = torch.zeros(Cc, device="cuda"), torch.zeros(Cc, device="cuda")
        seedoff = torch.zeros(1, dtype=torch.int64, device="cuda")
        drop = (L.DROP_RNG_ELEM, 0.1, 1234, None, seedoff)
        n = y.numel() * esz
        res = {}
        L.bn_fused_fwd(y, z, stats, gamma, beta, rmean, rvar, nbt, 0.1, 1e-5, True, coef, 0.01, drop)
        sc, sh, mu, istd = coef[0], coef[1], coef[2], coef[3]
        res["bn_fwd"] = (timeit(lambda: L.bn_fused_fwd(y, z, stats, gamma, beta, rmean, rvar, nbt, 0.1, 1e-5, True, coef,
                                                       0.01, drop)), 2 * n)
        res["bn_bwd_reduce"] = (timeit(lambda: L.bn_act_bwd_reduce(dz, y, sc, sh, mu, istd, sums, 0.01, drop)), 2 * n)
        res["bn_bwd_apply"] = (timeit(lambda: L.bn_act_bwd_apply(dz, y, sc, sh, mu, istd, sums, True, dy, dgam, dbet, 0.01,
                                                                 drop)), 3 * n)
        if H > 16:
            pz = torch.empty(N, H // 2, H // 2, Cc, device="cuda", dtype=td)
            dpz = torch.randn(N, H // 2, H // 2, Cc, device="cuda").to(td)
            res["maxpool_fwd"] = (timeit(lambda: L.maxpool2_fwd(z, pz)), 1.25 * n)
            res["maxpool_bwd"] = (timeit(lambda: L.maxpool2_bwd(z, dpz, dy)), 2.25 * n)
            # the level's up-sampling input has half the resolution and this level's channel count
            ux = torch.randn(N, H // 2, H // 2, Cc, device="cuda").to(td)
            res["upsample_fwd"] = (timeit(lambda: L.upsample2x_fwd(ux, z)), 1.25 * n)
            res["upsample_bwd"] = (timeit(lambda: L.upsample2x_bwd(dz, ux)), 1.25 * n)
        out[f"{H:3d}x{H:<3d} C={Cc:3d}"] = res
    print(repr(out))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("libs", nargs="*")
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--reps", type=int, default=30)
    ap.add_argument("--child", action="store_true")
    a = ap.parse_args()
    if a.child:
        child(a.dtype, a.reps)
        return
    res = {}
    for spec in (a.libs or [""]):
        path, _, envs = spec.partition("@")
        env = dict(os.environ)
        if path:
            env["FEDICRA_HIP_LIB"] = os.path.abspath(path)
        for kv in filter(None, envs.split(",")):
            k, v = kv.split("=")
            env[k] = v
        o = subprocess.run([sys.executable, __file__, "--child", "--dtype", a.dtype, "--reps", str(a.reps)], env=env,
                           capture_output=True, text=True)
        if o.returncode:
            print(spec, "FAILED", o.stderr[-600:])
            continue
        res[(os.path.basename(path).replace(".so", "") or "in-tree") + ("@" + envs if envs else "")] = ast.literal_eval(
            o.stdout.strip().splitlines()[-1])
    names = list(res)
    if not names:
        return
    ops = ["bn_fwd", "bn_bwd_reduce", "bn_bwd_apply", "maxpool_fwd", "maxpool_bwd", "upsample_fwd", "upsample_bwd"]
    print(f"{'level':16s} {'op':14s} " + " | ".join(f"{n[:24]:>24s}" for n in names) + "   (us, GB/s)")
    tot = {n: 0.0 for n in names}
    mult = {"256": 4, "128": 4, " 64": 4, " 32": 4, " 16": 2}         # BN layers per level; pools/upsample: one each
    for lvl in res[names[0]]:
        for op in ops:
            if op not in res[names[0]][lvl]:
                continue
            cells = []
            for n in names:
                us, nbytes = res[n][lvl][op]
                tot[n] += us * (mult[lvl[:3]] if op.startswith("bn_") else 1)
                cells.append(f"{us:8.1f} {nbytes / us / 1e3:8.0f}")
            print(f"{lvl:16s} {op:14s} " + " | ".join(f"{c:>24s}" for c in cells))
    print(f"{'per U-Net step':31s} " + " | ".join(f"{tot[n]:24.0f}" for n in names))


if __name__ == "__main__":
    main()
