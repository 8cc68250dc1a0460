import sys, torch
sys.path.insert(0, '.')
sys.path.insert(0, 'tests')
from fedicra_amd import _lib as L
import test_fused_gpu as T
DEV = "cuda"
for case in range(len(T.V2_CASES)):
    N, H, W, c0, c1, cout, G, kind, stats, two = T.V2_CASES[case]
    td = torch.bfloat16
    gen = torch.Generator().manual_seed(100 + case)
    pool, shared = kind == "pool", kind == "shared"
    B = N // G
    hs, ws_ = (2 * H, 2 * W) if pool else (H, W)
    x0 = torch.randn(B if shared else N, hs, ws_, c0, generator=gen).to(DEV).to(td)
    x1 = torch.randn(N, H, W, c1, generator=gen).to(DEV).to(td) if c1 else None
    w = (torch.randn(cout, 3, 3, c0 + c1, generator=gen) * 0.05).to(DEV).to(td)
    bias = torch.randn(cout, generator=gen).to(DEV)
    soff = torch.full((1,), 3, dtype=torch.int32, device=DEV)
    t0 = t1 = None
    if kind != "none":
        drop = (L.DROP_RNG_ELEM, 0.25, 0xABCDE, None, soff) if kind in ("drop", "shared") else None
        t0 = L.in_xform(T._coef(G, c0, gen), 0.01, pool=pool, drop=drop, seed_group_stride=0x10001)
        if c1:
            t1 = L.in_xform(T._coef(G, c1, gen), 0.0)
    if two or stats == "only":
        continue

    def run():
        y = torch.empty(N, H, W, cout, dtype=td, device=DEV)
        if kind == "none":
            L.conv2d_fwd(x0, x1, w, bias, y, None, None, ksize=3)
        else:
            L.conv2d_fwd_fused(x0, t0, x1, t1, w, bias, y, None, ksize=3, groups=G, cout=cout, shared0=shared)
        return y
    L.conv_tuning(0)
    want = run()
    for nf in (1, 2, 4):
        if nf > 1 and (nf // 2) * 16 >= cout:
            continue
        for ck in (16, 32):
            L.conv_tuning(1, nf, ck, 2)
            got = run()
            torch.cuda.synchronize()
            d = (got.float() - want.float()).abs()
            bad = d > 0
            if bad.any():
                idx = bad.nonzero()
                print(f"case {case} {T.V2_CASES[case]} nf{nf} ck{ck}: max {d.max().item():.3e} nbad {int(bad.sum())}/{bad.numel()} "
                      f"n {idx[:,0].unique().tolist()} rows {idx[:,1].min().item()}-{idx[:,1].max().item()} cols {idx[:,2].min().item()}-{idx[:,2].max().item()} "
                      f"ch {idx[:,3].min().item()}-{idx[:,3].max().item()}")
            else:
                print(f"case {case} nf{nf} ck{ck}: ok")
L.conv_tuning(-1)
