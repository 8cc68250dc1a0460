"""conv_fwd_ws2_kernel (64 x 64 wave tiles on v_mfma_f32_32x32x16, chunk-major weights; csrc/conv_ws2.h) against the one-tile
kernel on the same operands, and the chunk-major pack modes against their definition."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"
TD = {"bf16": torch.bfloat16, "fp16": torch.float16}


def _coef(G, C, gen):
    return torch.stack([torch.rand(G, C, generator=gen) + 0.5, torch.randn(G, C, generator=gen) * 0.3]).to(DEV)


@pytest.mark.parametrize("dtype", ["bf16", "fp16"])
def test_chunk_major_pack_modes_follow_their_definition(dtype):
    """fi_pack_weights modes 2 / 3 and the chunk-major columns of fi_pack_weights_multi: dst[ci/16][co][t][ci%16] of the forward
    operand, dst[co/16][ci][8-t][co%16] of the dgrad operand (include/fedicra_hip.h)."""
    from fedicra_amd import _lib as L
    td = TD[dtype]
    gen = torch.Generator().manual_seed(3)
    for cout, cin in ((64, 32), (128, 96), (64, 64)):
        wf = torch.randn(cout, 9, cin, generator=gen).to(DEV)
        std, cm, dg, dgcm = (torch.empty(cout * 9 * cin, dtype=td, device=DEV) for _ in range(4))
        L.pack_weights(wf, std, cout, 9, cin, 0)
        L.pack_weights(wf, cm, cout, 9, cin, 2)
        L.pack_weights(wf, dg, cout, 9, cin, 1)
        L.pack_weights(wf, dgcm, cout, 9, cin, 3)
        w = wf.to(td)
        assert torch.equal(std.view(cout, 9, cin), w)
        want_cm = w.view(cout, 9, cin // 16, 16).permute(2, 0, 1, 3).contiguous()
        assert torch.equal(cm.view(cin // 16, cout, 9, 16), want_cm)
        wt = w.flip(1).permute(2, 1, 0).contiguous()                      # [ci][8-t][co]: the dgrad operand
        assert torch.equal(dg.view(cin, 9, cout), wt)
        want_dgcm = wt.view(cin, 9, cout // 16, 16).permute(2, 0, 1, 3).contiguous()
        assert torch.equal(dgcm.view(cout // 16, cin, 9, 16), want_dgcm)
        # the multi-tensor form writes the same four operands
        o = [torch.empty(cout * 9 * cin, dtype=td, device=DEV) for _ in range(4)]
        table = torch.tensor([[wf.data_ptr(), o[0].data_ptr(), o[1].data_ptr(), cout, 9, cin, o[2].data_ptr(), o[3].data_ptr()]],
                             dtype=torch.int64, device=DEV)
        L.pack_weights_multi(table, 1, td)
        torch.cuda.synchronize()
        for a, b in zip(o, (std, dg, cm, dgcm)):
            assert torch.equal(a, b)
    assert L.conv_weight_chunk16(td, 3, 64, 128) and not L.conv_weight_chunk16(td, 3, 16, 128)
    assert not L.conv_weight_chunk16(td, 1, 64, 128) and not L.conv_weight_chunk16(torch.float32, 3, 64, 128)


WS2_CASES = [
    # (N, H, W, c0, c1, cout, groups, kind, stats, two_dst)
    (4, 32, 48, 64, 0, 128, 2, "drop", True, False),
    (4, 16, 16, 128, 128, 64, 1, "xf", False, False),        # two sources, 32-row tiles on a 16-row map
    (2, 64, 64, 64, 0, 512, 1, "xf", "only", False),         # the auxiliary head: statistics-only
    (3, 48, 32, 64, 0, 64, 1, "none", False, True),          # dgrad w.r.t. a concatenation: two destinations
    (8, 16, 16, 256, 0, 256, 4, "shared", True, False),      # one group's images read by every group, dropout per group
    (6, 33, 21, 32, 32, 64, 3, "xf", True, False),           # ragged tile edges, two sources
    (5, 37, 50, 32, 0, 128, 1, "none", True, False),         # plain loader, ragged, bias + statistics
    (2, 4, 4, 256, 0, 256, 1, "none", True, False),          # the 4 x 4 map of a 64^2 input
    (36, 128, 128, 32, 0, 128, 4, "drop", True, False),      # > 8 tiles per workgroup: the strip is flushed inside a run
    (3, 40, 24, 64, 64, 128, 1, "none", False, True),        # two sources AND two destinations
    # whole filter resident in LDS (forms 3 / 4: Cout = 64 / 32) -- cases 1, 3 and 5 above qualify as well
    (5, 70, 40, 32, 32, 32, 1, "xf", True, False),           # 32 outputs, two sources, ragged
    (8, 64, 64, 32, 0, 32, 4, "drop", True, False),
    (4, 40, 40, 128, 0, 32, 2, "shared", True, False),       # 8 resident chunks
    (40, 128, 128, 64, 0, 64, 4, "drop", True, False),       # > 8 tiles per workgroup
    (3, 24, 56, 64, 0, 32, 1, "none", True, False),          # plain loader, bias + statistics
    # one 64-channel pixel group, 2..4 slabs of 128: the slabs are walked inside a tile (form 5) -- case 2 above as well
    (6, 40, 24, 64, 0, 256, 3, "drop", True, False),
    (3, 33, 50, 64, 0, 384, 1, "none", True, False),         # plain loader, three slabs, ragged
    (40, 128, 128, 64, 0, 256, 4, "xf", True, False),        # > 8 tiles per workgroup: every slab's strip flushed inside a run
    (5, 48, 48, 64, 0, 512, 1, "shared", "only", False),     # statistics only, four slabs
]


@pytest.mark.parametrize("dtype", ["bf16", "fp16"])
@pytest.mark.parametrize("case", range(len(WS2_CASES)))
def test_ws2_kernel_equals_the_one_tile_kernel(dtype, case):
    """Same fp32 products; the channel chunks are accumulated 16 at a time (the one-tile kernel: 32 or 64), so outputs agree
    to the last bit or two of the storage type; statistics are those of the stored outputs."""
    from fedicra_amd import _lib as L
    N, H, W, c0, c1, cout, G, kind, stats, two = WS2_CASES[case]
    if dtype == "fp16" and case in (8, 13, 17):
        pytest.skip("the long-run case once is enough")
    td = TD[dtype]
    gen = torch.Generator().manual_seed(300 + case)
    shared = kind == "shared"
    B = N // G
    x0 = torch.randn(B if shared else N, H, W, c0, generator=gen).to(DEV).to(td)
    x1 = torch.randn(N, H, W, c1, generator=gen).to(DEV).to(td) if c1 else None
    wf = (torch.randn(cout, 3, 3, c0 + c1, generator=gen) * 0.05).to(DEV)
    w = wf.to(td)
    w16 = torch.empty(w.numel(), dtype=td, device=DEV)
    L.pack_weights(wf, w16, cout, 9, c0 + c1, 2)
    bias = torch.randn(cout, generator=gen).to(DEV)
    soff = torch.full((1,), 3, dtype=torch.int32, device=DEV)
    t0 = t1 = None
    if kind != "none":
        drop = (L.DROP_RNG_ELEM, 0.25, 0xABCDE, None, soff) if kind in ("drop", "shared") else None
        t0 = L.in_xform(_coef(G, c0, gen), 0.01, drop=drop, seed_group_stride=0x10001)
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

    ulp = 2.0 ** -7 if dtype == "bf16" else 2.0 ** -10
    try:
        L.conv_tuning(0)
        want, st_want = run()                                  # no chunk-major operand attached: the one-tile kernel
        w._fi_w16 = w16
        # 8 = the LDS-DMA GEMM tile (conv_fwd_dma_kernel, csrc/conv_dma.h): 128-channel slabs
        forms = ([1, 8] if cout % 128 == 0 else []) + ([2] if cout % 64 == 0 else [])
        if cout in (32, 64) and 78336 + (c0 + c1) * cout * 18 + 768 <= 160 * 1024:
            forms.append(4)
        assert forms
        for tr in forms:
            for wgs in (1, 8):
                L.conv_tuning(7, tr, 0, wgs)
                got, st_got = run()
                torch.cuda.synchronize()
                for a, b in zip(got, want):
                    if a is not None:
                        d = (a.float() - b.float()).abs()
                        assert bool((d <= 2 * ulp * b.float().abs() + 1e-3).all()), (tr, wgs, float(d.max()))
                        assert float((d > 0).float().mean()) < 4e-3, (tr, wgs, float((d > 0).float().mean()))
                if st_want is not None:
                    assert torch.allclose(st_got.sum(1), st_want.sum(1), rtol=5e-3, atol=1e-2), (tr, wgs)
    finally:
        L.conv_tuning(-1)
