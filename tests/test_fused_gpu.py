"""The fused ("probe") forward against the unfused kernels it replaces: the transforming conv loader (BatchNorm apply +
LeakyReLU + dropout [+ 2x2 max-pool] while staging) vs fi_bn_act_fwd [+ fi_maxpool2_fwd] + fi_conv2d_fwd, and the batched
K-1 no-grad forwards of FedICRA's LC loss vs the K-1 separate forwards (flower_pCE_2D.py:128-139)."""
import argparse

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"
TD = {"fp32": torch.float32, "bf16": torch.bfloat16, "fp16": torch.float16}


def _coef(G, C, gen):
    sc = (0.5 + torch.rand(G, C, generator=gen)).float()
    sh = (torch.randn(G, C, generator=gen) * 0.3).float()
    return torch.stack([sc, sh]).to(DEV).contiguous()          # [2][G][C]


@pytest.mark.parametrize("dtype", ["fp32", "bf16", "fp16"])
@pytest.mark.parametrize("case", ["plain", "drop", "pool", "cat", "k1", "shared", "wide", "ragged"])
def test_transforming_loader_equals_bn_act_then_conv(dtype, case):
    """Same bits out of the convolution whether its input z was materialised by fi_bn_act_fwd (and pooled by
    fi_maxpool2_fwd) or evaluated by the loader from the raw y; the statistics epilogue lands in the group's accumulators."""
    from fedicra_amd import _lib as L
    td = TD[dtype]
    gen = torch.Generator().manual_seed(len(case) * 31 + ord(case[0]))
    G, B = (3, 2)
    H = W = 24
    c0, c1, cout, k = 16, 0, 32, 3
    drop_p, pool, shared = 0.0, False, False
    if case == "drop":
        drop_p = 0.3
    elif case == "pool":
        pool = True
    elif case == "cat":
        c0, c1 = 16, 16
    elif case == "k1":
        k, c0, cout = 1, 32, 16
    elif case == "shared":
        shared, drop_p = True, 0.2
    elif case == "wide":
        c0, cout = 128, 64
    elif case == "ragged":
        H, W = 20, 27
    N = G * B
    hs, ws = (2 * H, 2 * W) if pool else (H, W)
    y0 = torch.randn(B if shared else N, hs, ws, c0, generator=gen).to(DEV).to(td)
    coef0 = _coef(G, c0, gen)
    x1 = torch.randn(N, H, W, c1, generator=gen).to(DEV).to(td) if c1 else None
    w = (torch.randn(cout, k, k, c0 + c1, generator=gen) * 0.1).to(DEV).to(td)
    bias = torch.randn(cout, generator=gen).to(DEV)
    soff = torch.full((1,), 5, dtype=torch.int32, device=DEV)
    seed, stride = 0x1234567, 0x10001
    # --- unfused: per group bn_act (+pool) then one conv over the whole batch
    z = torch.empty((N, hs, ws, c0), dtype=td, device=DEV)
    for g in range(G):
        src = y0 if shared else y0[g * B:(g + 1) * B]
        drop = (L.DROP_RNG_ELEM, drop_p, seed + g * stride, None, soff) if drop_p > 0 else None
        L.bn_act_fwd(src, coef0[0, g], coef0[1, g], z[g * B:(g + 1) * B], 0.01, drop)
    if pool:
        zp = torch.empty((N, H, W, c0), dtype=td, device=DEV)
        L.maxpool2_fwd(z, zp)
        z = zp
    want = torch.empty((N, H, W, cout), dtype=td, device=DEV)
    st_want = torch.zeros(G, L.STATS_SLOTS, cout, 2, dtype=torch.float64, device=DEV)
    for g in range(G):
        L.conv2d_fwd(z[g * B:(g + 1) * B], None if x1 is None else x1[g * B:(g + 1) * B], w, bias, want[g * B:(g + 1) * B],
                     None, st_want[g], ksize=k)
    # --- fused
    drop = (L.DROP_RNG_ELEM, drop_p, seed, None, soff) if drop_p > 0 else None
    t0 = L.in_xform(coef0, 0.01, pool=pool, drop=drop, seed_group_stride=stride)
    got = torch.empty_like(want)
    st_got = torch.zeros_like(st_want)
    L.conv2d_fwd_fused(y0, t0, x1, None, w, bias, got, st_got, ksize=k, groups=G, shared0=shared)
    torch.cuda.synchronize()
    assert torch.equal(got, want), float((got.float() - want.float()).abs().max())
    a, b = st_got.sum(1), st_want.sum(1)
    # fp32 partial sums per tile (one-tile kernel) or per lane over a workgroup's tiles (thin-layer kernel), fp64 across them
    assert torch.allclose(a, b, rtol=2e-6, atol=2e-3), float((a - b).abs().max())
    # statistics-only launch: nothing stored, same accumulators
    st_only = torch.zeros_like(st_want)
    L.conv2d_fwd_fused(y0, t0, x1, None, w, bias, None, st_only, ksize=k, groups=G, cout=cout, shared0=shared)
    assert torch.allclose(st_only.sum(1), b, rtol=2e-6, atol=2e-3)


def test_bn_finalize_groups_moves_the_running_statistics_like_consecutive_forwards():
    from fedicra_amd import _lib as L
    gen = torch.Generator().manual_seed(5)
    G, C, count = 4, 48, 1000.0
    s1 = torch.randn(G, L.STATS_SLOTS, C, generator=gen).double() * 30
    s2 = (torch.rand(G, L.STATS_SLOTS, C, generator=gen).double() + 1.0) * 400
    stats = torch.stack([s1, s2], dim=-1).to(DEV).contiguous()
    gamma, beta = torch.rand(C, generator=gen).to(DEV) + 0.5, torch.randn(C, generator=gen).to(DEV)
    rm0, rv0 = torch.randn(C, generator=gen).to(DEV), torch.rand(C, generator=gen).to(DEV) + 0.5
    rm, rv, nbt = rm0.clone(), rv0.clone(), torch.zeros(1, dtype=torch.int64, device=DEV)
    want = torch.empty(2, G, C, device=DEV)
    tmp = torch.empty(2, C, device=DEV)
    for g in range(G):
        L.bn_finalize(stats[g], count, gamma, beta, rm, rv, nbt, 0.1, 1e-5, True, want[0, g], want[1, g], tmp[0], tmp[1])
    rm2, rv2, nbt2 = rm0.clone(), rv0.clone(), torch.zeros(1, dtype=torch.int64, device=DEV)
    coef = torch.empty(2, G, C, device=DEV)
    L.bn_finalize_groups(stats, G, count, gamma, beta, rm2, rv2, nbt2, 0.1, 1e-5, coef)
    torch.cuda.synchronize()
    assert torch.equal(coef, want) and torch.equal(rm, rm2) and torch.equal(rv, rv2) and int(nbt2) == G == int(nbt)
    # shared accumulators: every group the same coefficients, running statistics still moved G times
    rm3, rv3, nbt3 = rm0.clone(), rv0.clone(), torch.zeros(1, dtype=torch.int64, device=DEV)
    L.bn_finalize_groups(stats[0].contiguous(), G, count, gamma, beta, rm3, rv3, nbt3, 0.1, 1e-5, coef, shared=True)
    rm4, rv4 = rm0.clone(), rv0.clone()
    for g in range(G):
        L.bn_finalize(stats[0], count, gamma, beta, rm4, rv4, nbt, 0.1, 1e-5, True, tmp[0], tmp[1], want[0, 0], want[1, 0])
    assert torch.equal(rm3, rm4) and torch.equal(rv3, rv4) and all(torch.equal(coef[:, g], coef[:, 0]) for g in range(G))


@pytest.mark.parametrize("dtype,model", [("fp32", "unet_lc"), ("bf16", "unet_lc"), ("bf16", "unet_lc_multihead")])
def test_batched_probe_forwards_equal_the_separate_forwards(dtype, model):
    """`probe_heatmaps(x, others)` against `model(x, j, heatmap_only=True)` for every other client j from the same state and
    dropout stream: the same heat-maps, the same BatchNorm running statistics and counters afterwards (statistics are
    fp64 atomic sums whose order differs between launch shapes: an fp32 ulp or two), the same dropout call counters."""
    from fedicra_amd import ops
    from fedicra_amd.networks import net_factory
    from fedicra_amd.networks.unet import set_compute_dtype
    from helpers import loader
    in_chns = 3 if model == "unet_lc" else 1
    b = loader(1, 4, 64, cid=1, in_chns=in_chns, ncls=in_chns, device=DEV)[0]["image"]
    x = b if in_chns == 3 else b.unsqueeze(1)
    K, cid = 5, 1
    args = argparse.Namespace(min_num_clients=K, cid=cid)
    others = [j for j in range(K) if j != cid]
    res = []
    for batched in (False, True):
        torch.manual_seed(2022)
        ops.manual_seed(3)
        net = net_factory(args, net_type=model, in_chns=in_chns, class_num=in_chns).cuda().train()
        set_compute_dtype(net, dtype)
        ctx = ops.new_context()
        ctx.seed_offset = torch.full((1,), 7, dtype=torch.int32, device=DEV)
        with ops.use_context(ctx), torch.no_grad():
            ops.begin_iteration(x.device)
            net(x)                                            # the iteration's own forward comes first (call counters at 1)
            if batched:
                hm = net.probe_heatmaps(x, others)
                assert hm is not None
            else:
                hm = [net(x, j, heatmap_only=True)[6][-1] for j in others]
        torch.cuda.synchronize()
        res.append(([h.float().clone() for h in hm], net.flat_state.clone(), net.flat_counters.clone(), dict(ctx.call_idx)))
    (h0, s0, c0, k0), (h1, s1, c1, k1) = res
    assert k0 == k1 and torch.equal(c0, c1) and int(c1.min()) == K
    tol = 1e-5 if dtype == "fp32" else 2e-2
    for a, b_ in zip(h0, h1):
        assert a.shape == b_.shape and torch.allclose(a, b_, rtol=0, atol=tol), float((a - b_).abs().max())
    assert len({float(h.sum()) for h in h1}) == len(others)   # different embeddings / masks: different maps
    assert torch.allclose(s0, s1, rtol=1e-5 if dtype == "fp32" else 5e-3, atol=1e-6 if dtype == "fp32" else 1e-3), \
        float((s0 - s1).abs().max())


def test_probe_heatmaps_declines_when_it_does_not_apply():
    from fedicra_amd import ops
    from fedicra_amd.networks import net_factory
    from helpers import loader
    x = loader(1, 2, 32, cid=0, device=DEV)[0]["image"].unsqueeze(1)
    net = net_factory(argparse.Namespace(min_num_clients=3, cid=0), net_type="unet_lc", in_chns=1, class_num=2).cuda().train()
    with torch.no_grad():
        with ops.use_context(ops.new_context()):              # (a fresh context: an earlier test's may carry a stream)
            assert net.probe_heatmaps(x, [1, 2]) is None      # no device RNG stream in this context
        ctx = ops.new_context()
        ctx.seed_offset = torch.zeros(1, dtype=torch.int32, device=DEV)
        with ops.use_context(ctx):
            ops.begin_iteration(x.device)
            assert net.probe_heatmaps(x, [1, 2]) is not None
            ops.set_dropout_mask_provider(lambda shape, p: torch.ones(shape))
            try:
                assert net.probe_heatmaps(x, [1, 2]) is None  # host-fed masks: separate forwards
            finally:
                ops.set_dropout_mask_provider(None)
            net.eval()
            assert net.probe_heatmaps(x, [1, 2]) is None
    net.train()
    ctx2 = ops.new_context()
    ctx2.seed_offset = torch.zeros(1, dtype=torch.int32, device=DEV)
    with ops.use_context(ctx2):
        assert net.probe_heatmaps(x, [1, 2]) is None          # autograd on


V2_CASES = [
    # (N, H, W, c0, c1, cout, groups, xf-kind, stats, two_dst)
    (4, 32, 32, 16, 0, 16, 1, "none", True, False),
    (4, 40, 56, 32, 0, 32, 2, "xf", True, False),
    (6, 33, 21, 16, 16, 16, 3, "xf", True, False),          # ragged tile edges, chunk spanning both sources
    (4, 32, 48, 64, 0, 128, 2, "drop", True, False),
    (4, 16, 16, 128, 128, 64, 1, "xf", False, False),
    (6, 24, 24, 16, 0, 32, 3, "pool", True, False),
    (2, 64, 64, 64, 0, 512, 1, "xf", "only", False),         # the auxiliary head: statistics-only
    (3, 48, 32, 64, 0, 64, 1, "none", False, True),          # dgrad w.r.t. a concatenation: two destinations
    (8, 16, 16, 256, 0, 256, 4, "shared", True, False),
]


@pytest.mark.parametrize("dtype", ["bf16", "fp16"])
@pytest.mark.parametrize("case", range(len(V2_CASES)))
def test_persistent_forward_kernel_equals_the_one_tile_kernel(dtype, case):
    """conv_fwd_v2_kernel (persistent workgroups, next stage in flight) against conv_fwd_kernel on the same operands, for every
    slab width / chunk / residency the host may pick: the same fp32 products; the channel chunks are accumulated in the
    order the chunk width gives, so outputs agree to the last bit or two of the storage type (and exactly wherever both
    kernels chunk alike)."""
    from fedicra_amd import _lib as L
    N, H, W, c0, c1, cout, G, kind, stats, two = V2_CASES[case]
    td = TD[dtype]
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
        t0 = L.in_xform(_coef(G, c0, gen), 0.01, pool=pool, drop=drop, seed_group_stride=0x10001)
        if c1:
            t1 = L.in_xform(_coef(G, c1, gen), 0.0)

    def run():
        st = torch.zeros(G, L.STATS_SLOTS, cout, 2, dtype=torch.float64, device=DEV) if stats else None
        if two:
            ya = torch.empty(N, H, W, cout // 2, dtype=td, device=DEV)
            yb = torch.empty(N, H, W, cout // 2, dtype=td, device=DEV)
            L.conv2d_fwd(x0, x1, w, None, ya, yb, None, ksize=3)
            return (ya, yb), None
        y = None if stats == "only" else torch.empty(N, H, W, cout, dtype=td, device=DEV)
        if kind == "none":
            L.conv2d_fwd(x0, x1, w, bias, y, None, None if st is None else st[0], ksize=3)
        else:
            L.conv2d_fwd_fused(x0, t0, x1, t1, w, bias, y, st, ksize=3, groups=G, cout=cout, shared0=shared)
        return (y,), st

    try:
        L.conv_tuning(0)
        want, st_want = run()
        nfs = [n for n in (1, 2, 4) if n == 1 or (n // 2) * 16 < cout]
        for nf in nfs:
            for ck in (16, 32):
                for wgs in (1, 3):
                    L.conv_tuning(1, nf, ck, wgs)
                    got, st_got = run()
                    torch.cuda.synchronize()
                    ulp = 2.0 ** -7 if dtype == "bf16" else 2.0 ** -10
                    for a, b in zip(got, want):
                        if a is not None:
                            d = (a.float() - b.float()).abs()
                            assert bool((d <= 2 * ulp * b.float().abs() + 1e-3).all()), (nf, ck, wgs, float(d.max()))
                            assert float((d > 0).float().mean()) < 2e-3, (nf, ck, wgs, float((d > 0).float().mean()))
                    if st_want is not None:
                        # statistics of the STORED outputs (a few of which differ in their last bit), per-tile fp32 partial sums grouped
                        # by the tile height each kernel picked
                        assert torch.allclose(st_got.sum(1), st_want.sum(1), rtol=5e-3, atol=1e-2), (nf, ck, wgs)
        if c0 + c1 <= 32 and cout <= 32 and not two:
            for wgs in (1, 4):                              # the thin-layer kernel (filter in registers, buffer loads / stores)
                L.conv_tuning(3, 0, 0, wgs)
                got, st_got = run()
                torch.cuda.synchronize()
                for a, b in zip(got, want):
                    if a is not None:
                        d = (a.float() - b.float()).abs()
                        assert bool((d <= 2 * ulp * b.float().abs() + 1e-3).all()), ("thin", wgs, float(d.max()))
                        assert float((d > 0).float().mean()) < 2e-3, ("thin", wgs)
                if st_want is not None:
                    assert torch.allclose(st_got.sum(1), st_want.sum(1), rtol=5e-3, atol=1e-2), ("thin", wgs)
        if c0 + c1 >= 32 and cout > 16:
            # the wave-specialised kernel (producer / consumer waves, double-buffered LDS stages), every slab width / chunk;
            # wgs = workgroups per CU of the persistent grid (1 = the product's; 8 = one item per workgroup at these sizes)
            for nf in ([2, 4] if cout > 32 else [2]):
                for ck in ([16] if pool else [16, 32]):
                    for form, wgs in ((4, 1), (5, 1), (5, 8), (6, 1), (6, 8)):       # 4 / 5 / 6 = 4 / 8 / 2 x 4 producer waves
                        L.conv_tuning(form, nf, ck, wgs)
                        got, st_got = run()
                        torch.cuda.synchronize()
                        for a, b in zip(got, want):
                            if a is not None:
                                d = (a.float() - b.float()).abs()
                                assert bool((d <= 2 * ulp * b.float().abs() + 1e-3).all()), ("ws", form, nf, ck, wgs, float(d.max()))
                                assert float((d > 0).float().mean()) < 2e-3, ("ws", form, nf, ck, wgs)
                        if st_want is not None:
                            assert torch.allclose(st_got.sum(1), st_want.sum(1), rtol=5e-3, atol=1e-2), ("ws", form, nf, ck, wgs)
    finally:
        L.conv_tuning(-1)


@pytest.mark.parametrize("dtype", ["bf16", "fp16", "fp32"])
def test_upsample_row_form_equals_the_flat_form_bit_for_bit(dtype):
    """fi_upsample2x_fwd picks the row form (one output row per workgroup, no divisions) for launches of >= 2^21 vectors -- the
    batched LC forwards -- and the flat form below that: the same interpolation expression, so one batched launch equals its
    images up-sampled one by one; and both agree with F.interpolate(align_corners=True)."""
    from fedicra_amd import _lib as L
    td = {"bf16": torch.bfloat16, "fp16": torch.float16, "fp32": torch.float32}[dtype]
    vg = 4 if dtype == "fp32" else 8
    for (N, h, w, C) in [(20, 128, 128, 4 * vg // 2), (12, 64, 96, 8 * vg), (5, 256, 256, 2 * vg)]:
        gen = torch.Generator().manual_seed(N * 1000 + C)
        x = torch.randn(N, h, w, C, generator=gen).to(DEV).to(td)
        assert N * 4 * h * w * (C // vg) >= (1 << 21) > 4 * h * w * (C // vg)         # batched: row form; one image: flat form
        y = torch.empty(N, 2 * h, 2 * w, C, dtype=td, device=DEV)
        L.upsample2x_fwd(x, y)
        y1 = torch.empty_like(y)
        for n in range(N):
            L.upsample2x_fwd(x[n:n + 1], y1[n:n + 1])
        torch.cuda.synchronize()
        assert torch.equal(y, y1), float((y.float() - y1.float()).abs().max())
        ref = torch.nn.functional.interpolate(x.float().permute(0, 3, 1, 2), scale_factor=2, mode="bilinear", align_corners=True)
        # fp32: the source coordinate o * (in-1)/(out-1) carries ~1e-7 * out of rounding on either side -> 1e-4 of a unit step
        tol = 3e-4 if dtype == "fp32" else 2e-2
        assert float((y.float().permute(0, 3, 1, 2) - ref).abs().max()) <= tol * max(1.0, float(ref.abs().max()))


@pytest.mark.parametrize("dtype", ["bf16", "fp16", "fp32"])
def test_split_global_pooling_equals_the_one_workgroup_form(dtype):
    """fi_global_avgmax_split (PCS pooling of the full-resolution maps: pixel ranges on separate workgroups, folded in range
    order) against the one-workgroup-per-image form and torch: maxima and FIRST arg-max identical (ties planted across range
    and lane boundaries), means to fp32 round-off of a different summation order; replays are bit-identical."""
    from fedicra_amd import _lib as L
    td = {"bf16": torch.bfloat16, "fp16": torch.float16, "fp32": torch.float32}[dtype]
    for (N, H, W, C) in [(12, 512, 512, 16), (3, 256, 384, 32), (5, 128, 128, 64)]:
        gen = torch.Generator().manual_seed(N + C)
        x = torch.randn(N, H, W, C, generator=gen).to(td)
        x[:, 3, 5, :] = 9.0
        x[:, H // 2, 7, :] = 9.0                               # the same maximum again, later in scan order
        x[:, H - 1, W - 1, : C // 2] = 9.0
        xd = x.to(DEV)
        S = L.lib().fi_global_avgmax_ranges(L.dt(td), N, H * W, C)
        assert S >= 2, (N, H, W, C)
        out = []
        for _ in range(2):
            avg, mx = torch.empty(N, C, device=DEV), torch.empty(N, C, device=DEV)
            am = torch.empty(N, C, dtype=torch.int32, device=DEV)
            L.global_avgmax(xd, avg, mx, am)                   # takes the split form at these sizes
            out.append((avg.clone(), mx.clone(), am.clone()))
        assert all(torch.equal(a, b) for a, b in zip(out[0], out[1]))
        avg1, mx1 = torch.empty(N, C, device=DEV), torch.empty(N, C, device=DEV)
        am1 = torch.empty(N, C, dtype=torch.int32, device=DEV)
        rc = L.lib().fi_global_avgmax(L.dt(td), L.ptr(xd), L.ptr(avg1), L.ptr(mx1), L.ptr(am1), N, H * W, C, L.stream())
        assert rc == 0
        torch.cuda.synchronize()
        avg, mx, am = out[0]
        assert torch.equal(mx, mx1) and torch.equal(am, am1)
        assert torch.equal(am.cpu().long(), torch.full((N, C), 3 * W + 5))
        ref = x.float().mean((1, 2))
        assert torch.allclose(avg.cpu(), ref, rtol=0, atol=2e-5) and torch.allclose(avg, avg1, rtol=0, atol=2e-5)
