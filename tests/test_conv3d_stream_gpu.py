"""conv3d_stream_kernel (csrc/conv3d_stream.hip: the thin full-resolution 3x3x3 layers of unet_3D streamed along the depth axis,
/root/reference/code/networks/unet_3D.py:40-41,60-61 with utils.py:99-123,260-276) against the general one-launch form on the same
operands (fi_conv3d_tuning switches between them in-process) and against torch's conv3d on the CPU."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"
TD = {"bf16": torch.bfloat16, "fp16": torch.float16}

# (N, D, H, W, c0, c1, cout0, cout1, kind)
CASES = [
    (2, 16, 32, 32, 16, 0, 16, 0, "fwd"),           # conv1.conv2 / up_concat1.conv2
    (1, 12, 40, 24, 16, 32, 16, 0, "fwd"),          # up_concat1.conv1: cat([skip 16, up-sampled 32]); ragged tiles
    (2, 9, 17, 33, 16, 0, 16, 0, "fwd"),            # odd depth, ragged rows and columns
    (1, 20, 32, 48, 16, 0, 16, 0, "dgrad"),         # input gradient 16 -> 16
    (2, 8, 24, 40, 16, 0, 16, 32, "dgrad"),         # input gradient of the concatenation: two destinations
    (1, 32, 64, 64, 16, 32, 16, 0, "fwd"),          # several runs per tile column
]


@pytest.mark.parametrize("dtype", ["bf16", "fp16"])
@pytest.mark.parametrize("case", range(len(CASES)))
def test_streaming_kernel_equals_the_general_form_and_torch(dtype, case):
    from fedicra_amd import _lib as L
    N, D, H, W, c0, c1, co0, co1, kind = CASES[case]
    td = TD[dtype]
    gen = torch.Generator().manual_seed(700 + case)
    cin, cout = c0 + c1, co0 + co1
    x0 = torch.randn(N, D, H, W, c0, generator=gen).to(DEV).to(td)
    x1 = torch.randn(N, D, H, W, c1, generator=gen).to(DEV).to(td) if c1 else None
    w = (torch.randn(cout, cin, 3, 3, 3, generator=gen) * 0.08).to(td)            # as the forward conv's [Cout][Cin][kD][kH][kW]
    # the library's operand: [Cout][kH * kW][kD][Cin] (ops3d._w_all mode 0; a dgrad's flipped / transposed filter has the same layout)
    w_all = w.permute(0, 3, 4, 2, 1).contiguous().view(cout, 9, 3, cin).to(DEV)
    bias = torch.randn(cout, generator=gen).to(DEV) if kind == "fwd" else None

    def run():
        st = torch.zeros(N, L.STATS_SLOTS, cout, 2, dtype=torch.float64, device=DEV) if kind == "fwd" else None
        if kind == "fwd":
            y = torch.empty(N, D, H, W, cout, dtype=td, device=DEV)
            assert L.conv3d_fwd_fused(x0, x1, w_all, bias, y, st, ksize=3)
            return (y,), st
        d0 = torch.empty(N, D, H, W, co0, dtype=td, device=DEV)
        d1 = torch.empty(N, D, H, W, co1, dtype=td, device=DEV) if co1 else None
        assert L.conv3d_dgrad_fused(x0, w_all, d0, d1, ksize=3)
        return (d0, d1), None

    try:
        L.conv3d_tuning(0)
        want, st_want = run()
        L.conv3d_tuning(1)
        got, st_got = run()
        torch.cuda.synchronize()
    finally:
        L.conv3d_tuning(-1)
    ulp = 2.0 ** -7 if dtype == "bf16" else 2.0 ** -10
    for a, b in zip(got, want):
        if a is not None:
            d = (a.float() - b.float()).abs()
            assert bool((d <= 2 * ulp * b.float().abs() + 2e-3).all()), float(d.max())
            assert float((d > 0).float().mean()) < 2e-2                            # same fp32 products, another accumulation order
    if st_want is not None:
        assert torch.allclose(st_got.sum(1), st_want.sum(1), rtol=5e-3, atol=2e-2)
    # torch on the CPU, fp32 arithmetic on the same 16-bit values
    xin = torch.cat([x0] + ([x1] if c1 else []), 4).float().cpu().permute(0, 4, 1, 2, 3)
    ref = torch.nn.functional.conv3d(xin, w.float(), None if bias is None else bias.cpu(), padding=1).permute(0, 2, 3, 4, 1)
    out = torch.cat([t.float().cpu() for t in got if t is not None], 4)
    err = (out - ref).abs()
    assert float(err.max()) <= 4 * ulp * float(ref.abs().max()) + 1e-3, float(err.max())
    if st_got is not None:
        s = st_got.sum(1).cpu()
        assert torch.allclose(s[..., 0], out.double().sum((1, 2, 3)), rtol=1e-4, atol=1e-2)
        assert torch.allclose(s[..., 1], (out.double() ** 2).sum((1, 2, 3)), rtol=1e-4, atol=1e-2)


@pytest.mark.gpu
def test_deferred_3x3x3_reduce_with_the_transposed_add_equals_the_permuting_reduce_and_numpy():
    """fi_wgrad_reduce_multi over rows whose word 9 is -cin (sums left in slice 0) + fi_wgrad_permute3d_multi, against the same table
    with word 9 = +cin (the reduce permutes as it adds) -- the same fixed-order sums, one add each: bit-equal -- and against the
    slices summed in float64 and permuted by numpy.  Ragged channel counts (48: a 64-chunk's tail; 80: two chunks), slice counts
    either side of the kernel's lane-group thresholds, a 2D row (word 9 = 0) in between."""
    import numpy as np
    from fedicra_amd import _lib as L
    shapes = [(16, 16, 3, True), (16, 48, 40, False), (24, 80, 17, True), (32, 0, 70, True), (256, 128, 2, False)]   # cout, cin3, slices, bias
    out = {}
    for sign in (1, -1):
        rows, keep, nblocks, nblocks3d = [], [], 0, 0
        for cout, cin3, slices, bias in shapes:
            n_dw = cout * 27 * cin3 if cin3 else cout * 9 * 32
            stride = (n_dw + cout + 3) & ~3
            g = torch.Generator().manual_seed(1000 + cout)
            part = torch.randn(slices, stride, generator=g).to(DEV)
            dw = torch.randn(n_dw, generator=g).to(DEV)
            db = torch.randn(cout, generator=g).to(DEV) if bias else None
            ll = 8 if slices <= 16 else 6 if slices <= 64 else 4
            rows.append([part.data_ptr(), stride, slices, dw.data_ptr(), n_dw, db.data_ptr() if bias else 0, cout, nblocks, ll,
                         sign * cin3, nblocks3d])
            nblocks += -(-stride // (4 << ll))
            if cin3 and sign < 0:
                nblocks3d += cout * -(-cin3 // 64)
            keep.append((part.clone(), dw.clone(), None if db is None else db.clone(), part, dw, db))
        assert len(rows[0]) == L.WGRAD_ROW
        table = torch.tensor(rows, dtype=torch.int64).to(DEV)
        L.wgrad_reduce_multi(table, len(rows), nblocks, nblocks3d)
        torch.cuda.synchronize()
        out[sign] = keep
    for (cout, cin3, slices, bias), a, b in zip(shapes, out[1], out[-1]):
        assert torch.equal(a[4], b[4]), (cout, cin3)
        if bias:
            assert torch.equal(a[5], b[5])
        part0, dw0, db0, _, dw, db = b
        n_dw = dw0.numel()
        s = part0.double().sum(0).cpu().numpy()
        w = s[:n_dw]
        if cin3:
            w = w.reshape(cout, 9, 3, cin3).transpose(0, 3, 2, 1).reshape(-1)              # [co][t][kd][ci] -> [co][ci][kd][t]
        np.testing.assert_allclose(dw.cpu().numpy(), dw0.cpu().numpy() + w, rtol=0, atol=2e-5 * max(1, slices) ** 0.5)
        if bias:
            np.testing.assert_allclose(db.cpu().numpy(), db0.cpu().numpy() + s[n_dw:n_dw + cout], rtol=0, atol=2e-5 * max(1, slices) ** 0.5)


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("shape", [(2, 8, 64, 64, 16, 0, 32), (1, 16, 64, 64, 32, 0, 32), (1, 26, 40, 64, 32, 64, 32),
                                   (2, 32, 32, 32, 32, 0, 64), (2, 32, 32, 32, 64, 128, 64)])
def test_row_streaming_3d_wgrad_channel_tiles_against_fp64_and_the_tile_kernels(shape, dtype):
    """conv_wgrad_rows3d_kernel with tiles of 2 gradient x 1 / 2 input blocks (unet_3D's 64^3 and 32^3 levels: 16 / 32 / 96 -> 32,
    32 / 64 / 192 -> 64) against an fp64 conv3d backward and the one-launch tile kernel it replaces: a pair split over two waves'
    K steps (16 -> 32), a pair a wave, several input / gradient tiles per item writing one partial slice, two sources with the
    boundary inside a tile, ragged row runs, 32-wide slices (one K step a row), depth padding at both ends of two volumes."""
    import torch.nn.functional as F
    from fedicra_amd import _lib as L
    N, D, H, W, c0, c1, cout = shape
    cin = c0 + c1
    g = torch.Generator().manual_seed(D + H + c1)
    x0 = torch.randn(N, D, H, W, c0, generator=g).to(dtype).to(DEV)
    x1 = torch.randn(N, D, H, W, c1, generator=g).to(dtype).to(DEV) if c1 else None
    dy = (torch.randn(N, D, H, W, cout, generator=g) * 0.1).to(dtype).to(DEV)
    res = []
    try:
        for rows in (3, 0):
            L.lib().fi_wgrad_tuning(rows)
            dw = torch.zeros(cout, 9, 3, cin, device=DEV)
            db = torch.zeros(cout, device=DEV)
            assert L.conv3d_wgrad_fused(x0, x1, dy, dw, db, ksize=3)
            res.append((dw.double().cpu(), db.double().cpu()))
    finally:
        L.lib().fi_wgrad_tuning(-1)
    x = x0 if x1 is None else torch.cat([x0, x1], 4)
    xd = x.double().permute(0, 4, 1, 2, 3)
    wref = torch.zeros(cout, cin, 3, 3, 3, dtype=torch.float64, device=DEV, requires_grad=True)
    F.conv3d(xd, wref, None, padding=1).backward(dy.double().permute(0, 4, 1, 2, 3))
    want = wref.grad.permute(0, 3, 4, 2, 1).reshape(cout, 9, 3, cin).cpu()          # [cout][kh][kw][kd][cin] -> [cout][9][3][cin]
    scale = want.abs().max().item()
    e_new = (res[0][0] - want).abs().max().item() / scale
    e_old = (res[1][0] - want).abs().max().item() / scale
    assert e_new < 2e-5 and e_old < 2e-5, (e_new, e_old)
    assert not torch.equal(res[0][0], res[1][0])               # another summation order: the row-streaming kernel really ran
    want_b = dy.double().sum((0, 1, 2, 3)).cpu()
    assert (res[0][1] - want_b).abs().max().item() < 2e-5 * max(1.0, want_b.abs().max().item())
