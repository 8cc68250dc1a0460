#!/usr/bin/env python
"""Kernel micro-benchmark: times individual C-ABI conv launches at the bench shapes (U-Net(1,2), 12x1x256x256)
for one or several builds of libfedicra_hip.so (A/B comparison of kernel variants on the same box, same data).

    python tools/kbench.py [--dtype bf16] [--reps 30] libA.so libB.so ...

Prints, per layer and kernel kind, the average microseconds per launch for every library given.
"""
import argparse
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


from fedicra_amd._lib import FiConv  # noqa: E402  (the C struct, incl. the chunk-major second operand fields)


# (H, c0, c1, cout, k) of every conv in UNet(1,2) at 256^2 (encoder, decoder), batch 12
LAYERS = [(256, 1, 0, 16, 3), (256, 16, 0, 16, 3), (128, 16, 0, 32, 3), (128, 32, 0, 32, 3), (64, 32, 0, 64, 3),
          (64, 64, 0, 64, 3), (32, 64, 0, 128, 3), (32, 128, 0, 128, 3), (16, 128, 0, 256, 3), (16, 256, 0, 256, 3),
          (16, 256, 0, 128, 1), (32, 128, 128, 128, 3), (32, 128, 0, 64, 1), (64, 64, 64, 64, 3), (64, 64, 0, 32, 1),
          (128, 32, 32, 32, 3), (128, 32, 0, 16, 1), (256, 16, 16, 16, 3), (256, 16, 0, 2, 3)]


def p(t):
    return C.c_void_p(0 if t is None else t.data_ptr())


def timeit(fn, reps):
    # captured into one hipGraph and replayed: the per-launch host overhead would hide anything below ~5 us
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


def bench_lib(path, dtype, reps, N=12):
    lib = C.CDLL(path)
    td = torch.bfloat16 if dtype == "bf16" else torch.float32
    di = 1 if dtype == "bf16" else 0
    has_ws = hasattr(lib, "fi_conv2d_wgrad_workspace")
    if has_ws:
        lib.fi_conv2d_wgrad_workspace.restype = C.c_long
    out = {}
    for (H, c0, c1, cout, k) in LAYERS:
        cin = c0 + c1
        x0 = torch.randn(N, H, H, c0, device="cuda").to(td)
        x1 = torch.randn(N, H, H, c1, device="cuda").to(td) if c1 else None
        w = (torch.randn(cout, k, k, cin, device="cuda") * 0.05).to(td)
        wt = (torch.randn(cin, k, k, cout, device="cuda") * 0.05).to(td)
        b = torch.randn(cout, device="cuda")
        y = torch.empty(N, H, H, cout, device="cuda", dtype=td)
        dy = torch.randn(N, H, H, cout, device="cuda").to(td)
        d0 = torch.empty(N, H, H, c0, device="cuda", dtype=td)
        d1 = torch.empty(N, H, H, c1, device="cuda", dtype=td) if c1 else None
        stats = torch.zeros(32 * cout * 2, dtype=torch.float64, device="cuda")
        dw = torch.zeros(cout, k, k, cin, device="cuda")
        db = torch.zeros(cout, device="cuda")
        dfw = FiConv(di, N, H, H, k, c0, c1, cout, 0, 0, 0, 0)
        ddg = FiConv(di, N, H, H, k, cout, 0, c0, c1, 0, 0, 0)
        key = f"{H:3d} {cin:3d}->{cout:3d} k{k}"

        def fwd():
            rc = lib.fi_conv2d_fwd(C.byref(dfw), p(x0), p(x1), p(w), p(b), p(y), None, p(stats), C.c_void_p(torch.cuda.current_stream().cuda_stream))
            assert rc == 0, rc

        def dgrad():
            rc = lib.fi_conv2d_fwd(C.byref(ddg), p(dy), None, p(wt), None, p(d0), p(d1), None, C.c_void_p(torch.cuda.current_stream().cuda_stream))
            assert rc == 0, rc

        if has_ws:
            nb = lib.fi_conv2d_wgrad_workspace(C.byref(dfw))
            ws = torch.empty(max(nb, 4) // 4, device="cuda")

            def wgrad():
                rc = lib.fi_conv2d_wgrad(C.byref(dfw), p(x0), p(x1), p(dy), p(dw), p(db), p(ws), C.c_long(nb), C.c_void_p(torch.cuda.current_stream().cuda_stream))
                assert rc == 0, rc
        else:
            def wgrad():
                rc = lib.fi_conv2d_wgrad(C.byref(dfw), p(x0), p(x1), p(dy), p(dw), p(db), C.c_void_p(torch.cuda.current_stream().cuda_stream))
                assert rc == 0, rc

        out[key] = (timeit(fwd, reps), timeit(dgrad, reps) if c0 > 1 else 0.0, timeit(wgrad, reps))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("libs", nargs="+")
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--reps", type=int, default=30)
    ap.add_argument("--child", action="store_true")
    a = ap.parse_args()
    if a.child:                                   # one library, results as a python literal on stdout
        print(repr(bench_lib(a.libs[0], a.dtype, a.reps)))
        return
    import ast
    import subprocess
    res = {}
    for spec in a.libs:                           # "path.so" or "path.so@ENV=VAL,ENV2=VAL2": env is read once per process
        path, _, envs = spec.partition("@")
        env = dict(os.environ)
        for kv in filter(None, envs.split(",")):
            k, v = kv.split("=")
            env[k] = v
        o = subprocess.run([sys.executable, __file__, "--child", "--dtype", a.dtype, "--reps", str(a.reps), path],
                           env=env, capture_output=True, text=True)
        if o.returncode:
            print(spec, "FAILED", o.stderr[-400:])
            continue
        res[os.path.basename(path).replace(".so", "") + ("@" + envs if envs else "")] = ast.literal_eval(
            o.stdout.strip().splitlines()[-1])
    names = list(res)
    print(f"{'layer':20s} " + " | ".join(f"{n[:22]:>22s}" for n in names) + "   (fwd / dgrad / wgrad, us)")
    tot = {n: [0.0, 0.0, 0.0] for n in names}
    for key in res[names[0]]:
        cells = []
        for n in names:
            f, d, w = res[n][key]
            for i, v in enumerate((f, d, w)):
                tot[n][i] += v
            cells.append(f"{f:6.1f} {d:6.1f} {w:6.1f}  ")
        print(f"{key:20s} " + " | ".join(f"{c:>22s}" for c in cells))
    print(f"{'TOTAL':20s} " + " | ".join(f"{tot[n][0]:6.0f} {tot[n][1]:6.0f} {tot[n][2]:6.0f}  ".rjust(22) for n in names))


if __name__ == "__main__":
    main()
