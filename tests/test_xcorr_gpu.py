"""csrc/xcorr.hip -- the statistics of a 3x3 convolution from the input's autocorrelation (the auxiliary head of FedICRA's LC
forwards, /root/reference/code/networks/unet.py:261-267, whose output nobody reads) -- against (a) an fp64 restatement of
its pieces in torch, (b) the library's own direct statistics-only launch, (c) the module path (BatchNorm running statistics
after UNet_LC.probe_heatmaps with and without it)."""
import ctypes as C

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda"
TAPS = [(0, 0), (0, 1), (0, 2), (1, -2), (1, -1), (1, 0), (1, 1), (1, 2), (2, -2), (2, -1), (2, 0), (2, 1), (2, 2)]


def _case(N, H, W, cout, groups, dtype, seed, xform=True):
    from fedicra_amd import _lib as L
    g = torch.Generator().manual_seed(seed)
    y = (torch.randn(N, H, W, 64, generator=g) * 1.3 + 0.2).to(dtype).to(DEV)
    coef = None
    if xform:
        coef = torch.stack([torch.rand(groups, 64, generator=g) + 0.5, torch.randn(groups, 64, generator=g) * 0.3]).to(DEV)
    w = (torch.randn(cout, 3, 3, 64, generator=g) * 0.05).to(DEV)
    bias = (torch.randn(cout, generator=g) * 0.1).to(DEV)
    wp = torch.empty(cout * 9 * 64, dtype=dtype, device=DEV)
    L.pack_weights(w, wp, cout, 9, 64, 0)
    return y, coef, w, bias, wp


def _z(y, coef, groups, slope):
    """The activation the loader builds: leaky(scale * y + shift), rounded to the storage dtype -> fp64 NHWC."""
    if coef is None:
        return y.double()
    N = y.shape[0]
    gi = N // groups
    sc = coef[0].repeat_interleave(gi, 0)[:, None, None, :]
    sh = coef[1].repeat_interleave(gi, 0)[:, None, None, :]
    t = torch.addcmul(sh, y.float(), sc)              # one rounding, like the kernel's contracted multiply-add
    t = torch.maximum(t, t * slope)
    return t.to(y.dtype).double()


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("shape", [(4, 12, 64, 2), (6, 37, 128, 3), (2, 5, 64, 1)])
def test_autocorrelation_matrices_and_channel_sums(shape, dtype):
    """A[g][d] = sum_q z(q) z(q + d)^T over the pixels where both lie inside an image, T[g] = sum_q z(q): the MFMA kernel
    (rows split over 36+ workgroups per group, image boundaries inside a workgroup's range, the two zero rows behind an
    image, the zero pad columns) against shifted fp64 products."""
    from fedicra_amd import _lib as L
    N, H, W, groups = shape
    y, coef, w, bias, wp = _case(N, H, W, 64, groups, dtype, 5)
    slope = 0.01
    stats = torch.zeros(groups * L.STATS_SLOTS * 64 * 2, dtype=torch.float64, device=DEV)
    d = L.FiConv(L.dt(dtype), N, H, W, 3, 64, 0, 64, 0, 0, 0, 0)
    need = L.lib().fi_conv2d_stats_xcorr_workspace(C.byref(d), N // groups)
    assert need > 0
    ws = torch.zeros(need, dtype=torch.uint8, device=DEV)
    assert L.conv2d_stats_xcorr(y, L.in_xform(coef, slope), wp, bias, stats, groups=groups, cout=64, workspace=ws)
    offs = (C.c_long * 8)()
    assert L.lib().fi_conv2d_stats_xcorr_layout(C.byref(d), N // groups, offs) == 0
    per = 13 * 4096 + 64
    A = ws[offs[1]:offs[1] + groups * per * 8].view(torch.float64).view(groups, per).cpu()
    z = _z(y, coef, groups, slope).cpu()
    gi = N // groups
    for g in range(groups):
        zg = z[g * gi:(g + 1) * gi]
        for k, (dr, dc) in enumerate(TAPS):
            a = zg[:, :H - dr, max(0, -dc):W - max(0, dc), :]
            b = zg[:, dr:, max(0, dc):W + min(0, dc), :]
            want = torch.einsum("nhwi,nhwj->ij", a, b)
            got = A[g, k * 4096:(k + 1) * 4096].view(64, 64)
            err = (got - want).abs().max().item() / max(1.0, want.abs().max().item())
            assert err < 2e-5, (g, (dr, dc), err)
        T = A[g, 13 * 4096:]
        assert ((T - zg.sum((0, 1, 2))).abs().max() / zg.abs().sum((0, 1, 2)).max()).item() < 1e-6


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("shape,cout,xform", [((4, 12, 64, 2), 72, True), ((6, 37, 128, 3), 512, True), ((3, 16, 128, 1), 256, False),
                                              ((8, 9, 64, 8), 64, True)])
def test_statistics_against_an_fp64_convolution_and_the_direct_launch(shape, cout, xform, dtype):
    """(sum, sum of squares) per group and output channel of y = conv3x3(z) + bias: the autocorrelation form against an fp64
    convolution of the same rounded operands (1e-6 relative: it IS the exact statistic) and against the direct
    statistics-only launch, which takes them of the 16-bit-rounded outputs (rounding noise: 2e-3 / sqrt(#pixels) on the sum,
    ~1e-5 on the sum of squares)."""
    from fedicra_amd import _lib as L
    N, H, W, groups = shape
    y, coef, w, bias, wp = _case(N, H, W, cout, groups, dtype, 9, xform)
    slope = 0.0
    t0 = L.in_xform(coef, slope)
    stats = torch.zeros(groups * L.STATS_SLOTS * cout * 2, dtype=torch.float64, device=DEV)
    assert L.conv2d_stats_xcorr(y, t0, wp, bias, stats, groups=groups, cout=cout)
    direct = torch.zeros_like(stats)
    L.conv2d_fwd_fused(y, t0, None, None, wp, bias, None, direct, ksize=3, groups=groups, cout=cout)
    got = stats.view(groups, L.STATS_SLOTS, cout, 2).sum(1).cpu()
    dir_ = direct.view(groups, L.STATS_SLOTS, cout, 2).sum(1).cpu()
    z = _z(y, coef, groups, slope)
    wq = wp.view(cout, 3, 3, 64).double().permute(0, 3, 1, 2)
    ref = F.conv2d(z.permute(0, 3, 1, 2), wq, bias.double(), padding=1)          # [N, cout, H, W] fp64 on the device
    gi = N // groups
    for g in range(groups):
        r = ref[g * gi:(g + 1) * gi]
        s1, s2 = r.sum((0, 2, 3)).cpu(), (r * r).sum((0, 2, 3)).cpu()
        scale1 = r.abs().sum((0, 2, 3)).cpu()
        # what is taken off again for the frame around the images comes from the library's 1x1 convolution, whose statistics
        # are those of 16-bit-rounded outputs: rounding noise of the ~2 (H + W) frame positions per image, nothing else
        n = float(gi * H * W)
        n_ring, rms = float(gi * 2 * (H + W + 2)), (s2 / n).sqrt()
        eps = (2.0 ** -9 if dtype == torch.bfloat16 else 2.0 ** -12) / 3 ** 0.5
        tol1 = 2e-6 * scale1 + 6.0 * rms * eps * n_ring ** 0.5
        tol2 = 2e-6 * s2 + 6.0 * 2.0 * 3.0 * rms * rms * eps * n_ring ** 0.5
        assert bool(((got[g, :, 0] - s1).abs() <= tol1).all()), (g, ((got[g, :, 0] - s1).abs() / tol1).max().item())
        assert bool(((got[g, :, 1] - s2).abs() <= tol2).all()), (g, ((got[g, :, 1] - s2).abs() / tol2).max().item())
        # the direct launch: rounding noise of ALL n outputs (6 sigma; sum of squares: d(y^2) = 2 y^2 delta, E y^4 ~ 3 rms^4)
        assert bool(((dir_[g, :, 0] - s1).abs() <= 2e-6 * scale1 + 6.0 * rms * eps * n ** 0.5).all()), g
        assert bool(((dir_[g, :, 1] - s2).abs() <= 2e-6 * s2 + 6.0 * 2.0 * 3.0 * rms * rms * eps * n ** 0.5).all()), g


def test_head_statistics_at_the_lc_forwards_shape_and_rate():
    """84 x 128^2 x 64 -> 512 in 7 groups, the launch of every FedICRA iteration: equal to the direct launch to its rounding
    noise; both timed (printed)."""
    from fedicra_amd import _lib as L
    N, H, W, groups, cout = 84, 128, 128, 7, 512
    y, coef, w, bias, wp = _case(N, H, W, cout, groups, torch.bfloat16, 3)
    out16 = torch.empty_like(wp)
    L.pack_weights(w, out16, cout, 9, 64, 2)
    wp._fi_w16 = out16
    t0 = L.in_xform(coef, 0.01)
    stats = torch.zeros(groups * L.STATS_SLOTS * cout * 2, dtype=torch.float64, device=DEV)
    direct = torch.zeros_like(stats)

    def timed(fn, reps=5):
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps * 1e3

    us_x = timed(lambda: L.conv2d_stats_xcorr(y, t0, wp, bias, stats.zero_(), groups=groups, cout=cout))
    us_d = timed(lambda: L.conv2d_fwd_fused(y, t0, None, None, wp, bias, None, direct.zero_(), ksize=3, groups=groups, cout=cout))
    a = stats.view(groups, L.STATS_SLOTS, cout, 2).sum(1)
    b = direct.view(groups, L.STATS_SLOTS, cout, 2).sum(1)
    n = float(N // groups * H * W)
    mean_a, mean_b = a[..., 0] / n, b[..., 0] / n
    var_a, var_b = a[..., 1] / n - mean_a ** 2, b[..., 1] / n - mean_b ** 2
    e_mean = ((mean_a - mean_b).abs() / var_b.sqrt()).max().item()
    e_var = ((var_a - var_b).abs() / var_b).max().item()
    print(f"head statistics 84x128^2 64->512: autocorrelation form {us_x:.1f} us, direct launch {us_d:.1f} us; "
          f"|d mean| / std {e_mean:.2e}, |d var| / var {e_var:.2e}")
    # the two differ by the direct launch's rounding of its 196 608 outputs per channel and group (bf16: 1.1e-3 relative each)
    assert e_mean < 1e-4 and e_var < 2e-4


@pytest.mark.parametrize("dtype", ["bf16", "fp16"])
def test_probe_heatmaps_moves_the_heads_running_statistics_as_the_direct_launch_does(dtype):
    """UNet_LC.probe_heatmaps (the K-1 LC forwards of a FedICRA iteration as one batched pass) with the auxiliary head's
    statistics from the autocorrelation form (256^2 input: the head reads 64 x 64 maps) against the same pass with the direct
    statistics-only launch (ops._XCORR off): heat-maps and every other tensor of the state bit for bit, the head's
    BatchNorm running mean / variance to the direct launch's rounding noise (its statistics are those of 16-bit-rounded
    outputs: 1.1e-3 / 1.4e-4 relative each, 4096 x 2 per channel and group here)."""
    import argparse
    from fedicra_amd import ops
    from fedicra_amd.networks import net_factory
    from fedicra_amd.networks.unet import set_compute_dtype
    from helpers import loader
    b = loader(1, 2, 256, cid=1, in_chns=3, ncls=3, device=DEV)[0]["image"]
    K, cid = 4, 1
    args = argparse.Namespace(min_num_clients=K, cid=cid)
    others = [j for j in range(K) if j != cid]
    res = []
    was = ops._XCORR
    try:
        for on in (False, True):
            ops._XCORR = on
            torch.manual_seed(2022)
            ops.manual_seed(3)
            net = net_factory(args, net_type="unet_lc", in_chns=3, class_num=3).cuda().train()
            set_compute_dtype(net, dtype)
            ctx = ops.new_context()
            ctx.seed_offset = torch.full((1,), 7, dtype=torch.int32, device=DEV)
            with ops.use_context(ctx), torch.no_grad():
                ops.begin_iteration(b.device)
                net(b)
                hm = net.probe_heatmaps(b, others)
                assert hm is not None
            torch.cuda.synchronize()
            res.append(([h.float().clone() for h in hm], {k: v.clone() for k, v in net.state_dict().items()}))
    finally:
        ops._XCORR = was
    (h0, s0), (h1, s1) = res
    for a, c in zip(h0, h1):
        assert torch.equal(a, c)
    moved = 0
    for k in s0:
        if "dsn_head.1.running" in k:
            ref = s0[k].float()
            scale = s0["decoder.dsn_head.1.running_var"].float().sqrt() if k.endswith("mean") else ref
            err = ((s1[k].float() - ref).abs() / scale).max().item()
            moved += int(not torch.equal(s1[k], s0[k]))
            assert err < (3e-4 if dtype == "bf16" else 5e-5), (k, err)
        else:
            assert torch.equal(s0[k], s1[k]), k
    assert moved >= 1                                  # the other form really ran (its values differ in the last bits)
