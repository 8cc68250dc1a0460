"""GPU parity of the 3D U-Net surface: the 3D pooling / trilinear kernels and the
depth-sliced Conv3d + InstanceNorm3d against torch's CPU functional ops on the same seeded inputs, and the whole
unet_3D against vectors produced by the reference's own module (tests/golden/g9_unet3d.npz).
Tolerances: fp32 mode -- logits within 1e-4; bf16 storage -- loose, reported."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda"
DTYPES = [torch.float32, torch.bfloat16, torch.float16]


def rnd(*shape, seed=0):
    g = torch.Generator().manual_seed(seed)
    return torch.rand(*shape, generator=g) * 2 - 1


def ndhwc(t, dtype):
    return t.permute(0, 2, 3, 4, 1).contiguous().to(dtype).to(DEV)


def ncdhw(t):
    return t.float().cpu().permute(0, 4, 1, 2, 3)


def close(got, ref, dtype, what, scale=1.0):
    atol = (1e-5 if dtype == torch.float32 else 3e-2) * scale * max(1.0, ref.abs().max().item())
    err = (got - ref).abs().max().item()
    assert err <= atol, f"{what}: max err {err:.3e} (atol {atol:.1e})"


@pytest.mark.parametrize("dtype", DTYPES)
def test_maxpool3d_and_trilinear_against_torch(dtype):
    from fedicra_amd import ops3d
    N, C, D, H, W = 2, 8, 4, 6, 8
    x = rnd(N, C, D, H, W, seed=1).to(dtype).float()
    x[:, :, ::2, ::2, ::2] = 0.25
    x[:, :, 1::2, 1::2, 1::2] = 0.25                     # ties across the window: the first in (d,h,w) order wins
    xr = x.clone().requires_grad_(True)
    yr = F.max_pool3d(xr, 2)
    g = rnd(*yr.shape, seed=2).to(dtype).float()
    yr.backward(g)
    xd = ndhwc(x, dtype).requires_grad_(True)
    yd = ops3d.maxpool3d(xd)
    yd.backward(ndhwc(g, dtype))
    close(ncdhw(yd.detach()), yr.detach(), dtype, "maxpool3d fwd")
    close(ncdhw(xd.grad), xr.grad, dtype, "maxpool3d bwd")
    for (d, h, w) in [(2, 3, 4), (1, 1, 1), (4, 4, 4), (3, 5, 2)]:
        u = rnd(N, C, d, h, w, seed=3).to(dtype).float().requires_grad_(True)
        up = F.interpolate(u, scale_factor=(2, 2, 2), mode="trilinear", align_corners=False)
        gg = rnd(*up.shape, seed=4).to(dtype).float()
        up.backward(gg)
        ud = ndhwc(u.detach(), dtype).requires_grad_(True)
        upd = ops3d.upsample3d2x(ud)
        upd.backward(ndhwc(gg, dtype))
        close(ncdhw(upd.detach()), up.detach(), dtype, f"trilinear fwd {d}x{h}x{w}")
        close(ncdhw(ud.grad), u.grad, dtype, f"trilinear bwd {d}x{h}x{w}", scale=4.0)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("case", [(2, 1, 0, 8, 6, 8, 8), (1, 8, 8, 16, 4, 8, 16), (2, 16, 0, 8, 2, 4, 4)])
def test_conv3d_instancenorm_relu_against_torch(dtype, case):
    """[Conv3d 3^3 -> InstanceNorm3d -> ReLU] on cat(x0, x1): output, input gradients, weight / bias gradients."""
    from fedicra_amd import ops3d
    N, c0, c1, cout, D, H, W = case
    cin = c0 + c1
    conv = torch.nn.Conv3d(cin, cout, 3, 1, 1)
    with torch.no_grad():
        conv.weight.copy_(rnd(*conv.weight.shape, seed=5) * (1.0 / (27 * cin) ** 0.5) * 1.7)
        conv.bias.copy_(rnd(cout, seed=6) * 0.1)
    x0 = rnd(N, c0, D, H, W, seed=7).to(dtype).float()
    x1 = rnd(N, c1, D, H, W, seed=8).to(dtype).float() if c1 else None
    wq = conv.weight.detach().to(dtype).float()
    xa = x0.clone().requires_grad_(True)
    xb = x1.clone().requires_grad_(True) if c1 else None
    wr = wq.clone().requires_grad_(True)
    br = conv.bias.detach().clone().requires_grad_(True)
    y = F.conv3d(torch.cat([xa, xb], 1) if c1 else xa, wr, br, padding=1)
    z = F.relu(F.instance_norm(y, eps=1e-5))
    g = rnd(*z.shape, seed=9).to(dtype).float()
    z.backward(g)
    conv = conv.to(DEV)
    with torch.no_grad():
        conv.weight.copy_(wq.to(DEV))
    d0 = ndhwc(x0, dtype).requires_grad_(c0 > 1)
    d1 = ndhwc(x1, dtype).requires_grad_(True) if c1 else None
    zd = ops3d.conv3d(d0, d1, conv, norm=True)
    zd.backward(ndhwc(g, dtype))
    close(ncdhw(zd.detach()), z.detach(), dtype, "conv3d+IN+ReLU fwd")
    if c0 > 1:
        close(ncdhw(d0.grad), xa.grad, dtype, "dx0", scale=4.0)
    if c1:
        close(ncdhw(d1.grad), xb.grad, dtype, "dx1", scale=4.0)
    close(conv.weight.grad.float().cpu(), wr.grad, dtype, "dW", scale=8.0)
    # the bias in front of an InstanceNorm has true gradient 0: both sides return round-off
    assert conv.bias.grad.abs().max().item() < (1e-3 if dtype == torch.float32 else 0.5)


def _mk(n_classes, in_channels, seed, dtype="fp32"):
    from fedicra_amd.networks.unet_3D import unet_3D
    from oracle.unet_ref import seeded_state
    m = unet_3D(n_classes=n_classes, in_channels=in_channels)
    seeded_state(m, seed)
    return m.cuda().set_compute_dtype(dtype)


def test_unet3d_logits_loss_and_gradients_match_reference_golden(golden):
    from helpers import assert_ck
    g = golden("g9_unet3d.npz")
    m = _mk(2, 1, 2031).eval()
    assert list(m.state_dict().keys()) == [str(k) for k in g["keys"]]
    x = torch.from_numpy(g["x"]).to(DEV)
    out = m(x)
    assert out.shape == (1, 2, 32, 32, 32) and out.dtype == torch.float32
    err = (out.detach().cpu() - torch.from_numpy(g["eval_logits"])).abs().max().item()
    assert err < 1e-4, f"3D eval logits max err {err:.3e}"
    lab = torch.from_numpy(g["labels"].astype(np.int64)).to(DEV)
    loss = F.cross_entropy(out, lab, ignore_index=2)              # loss glue only; every network op is a HIP launch
    assert abs(loss.item() - float(g["loss"])) < 1e-5
    loss.backward()
    gw = m.final.weight.grad.cpu()
    ref = torch.from_numpy(g["grad/final.weight"])
    assert (gw - ref).abs().max().item() <= 2e-4 * ref.abs().max().item() + 1e-7
    g1 = m.conv1.conv1[0].weight.grad.cpu()
    ref1 = torch.from_numpy(g["grad/conv1.conv1.0.weight"])
    assert (g1 - ref1).abs().max().item() <= 2e-3 * ref1.abs().max().item() + 1e-7
    for k, p in m.named_parameters():
        if k.endswith("weight"):
            assert_ck(p.grad.float().cpu(), g["grad_ck/" + k], rtol=5e-3, atol=1e-6, what="grad " + k)
    # two samples, three input channels, three classes
    m3 = _mk(3, 3, 2032).eval()
    x3 = torch.from_numpy(np.random.default_rng(910).random((2, 3, 32, 32, 32), dtype=np.float32)).to(DEV)
    with torch.no_grad():
        o3 = m3(x3)
    blk = o3[:, :, 10:14, 8:16, 8:16].cpu().numpy()
    assert np.abs(blk - g["eval_logits3_block"]).max() < 1e-4
    assert_ck(o3.cpu(), g["eval_logits3_ck"], rtol=2e-5, atol=1e-5, what="3-channel logits")


def test_unet3d_bf16_and_train_mode_dropout(golden):
    g = golden("g9_unet3d.npz")
    x = torch.from_numpy(g["x"]).to(DEV)
    ref = torch.from_numpy(g["eval_logits"])
    m = _mk(2, 1, 2031, dtype="bf16").eval()
    with torch.no_grad():
        o = m(x).cpu()
    agree = (o.argmax(1) == ref.argmax(1)).float().mean().item()
    assert agree > 0.97 and (o - ref).abs().max().item() < 0.25, (agree, (o - ref).abs().max().item())
    # train mode: dropout(0.3) after the centre block and before the final conv draws a fresh mask per call and the
    # backward pass regenerates it (finite gradients, different outputs)
    from fedicra_amd import ops
    m32 = _mk(2, 1, 2031).train()
    ops.manual_seed(3)
    a = m32(x)
    b = m32(x)
    assert not torch.equal(a, b)
    a.square().mean().backward()
    assert all(torch.isfinite(p.grad).all().item() for p in m32.parameters())


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("case", [(2, 16, 0, 16, 5, 21, 19), (1, 24, 8, 40, 3, 16, 33), (3, 64, 32, 64, 2, 9, 16)])
def test_one_launch_conv3d_equals_the_per_tap_form(dtype, case):
    """fi_conv3d_fwd_fused / dgrad_fused (one implicit GEMM over all slices, depth taps as channel groups, output written once)
    against fi_conv3d_fwd / dgrad (one launch per depth tap accumulating into the volume) on ragged volumes, two sources and
    two gradient destinations: the same products, fp32-accumulated in a different grouping and rounded to 16 bits once
    instead of three times -- they agree to the storage type's resolution, and the per-volume statistics to the accumulated rounding of their terms."""
    from fedicra_amd import _lib as L
    from fedicra_amd import ops3d
    N, c0, c1, cout, D, H, W = case
    cin = c0 + c1
    w = (rnd(cout, cin, 3, 3, 3, seed=31) * (1.7 / (27 * cin) ** 0.5)).to(DEV)
    bias = (rnd(cout, seed=32) * 0.1).to(DEV)
    x0 = rnd(N, D, H, W, c0, seed=33).to(dtype).to(DEV)
    x1 = rnd(N, D, H, W, c1, seed=34).to(dtype).to(DEV) if c1 else None
    wp = ops3d._w_taps(w, dtype, 0)
    w_all = torch.stack([t.view(cout, 9, cin) for t in wp], dim=2).contiguous()
    y0 = torch.zeros(N, D, H, W, cout, dtype=dtype, device=DEV)
    s0 = torch.zeros(N, L.STATS_SLOTS * cout * 2, dtype=torch.float64, device=DEV)
    L.conv3d_fwd(x0, x1, wp, bias, y0, s0, ksize=3)
    y1 = torch.full((N, D, H, W, cout), float("nan"), dtype=dtype, device=DEV)          # written, not accumulated
    s1 = torch.zeros_like(s0)
    assert L.conv3d_fwd_fused(x0, x1, w_all, bias, y1, s1, ksize=3)
    torch.cuda.synchronize()
    ulp = 2.0 ** -7 if dtype == torch.bfloat16 else 2.0 ** -10
    d = (y1.float() - y0.float()).abs()
    # (the per-tap form rounds its running sum after every tap: up to an ulp of the LARGEST partial sum, which may exceed |y|)
    assert bool((d <= 4 * ulp * (y0.float().abs() + 1.0)).all()), float(d.max())
    t0 = s0.view(N, L.STATS_SLOTS, cout, 2).sum(1)
    t1 = s1.view(N, L.STATS_SLOTS, cout, 2).sum(1)
    # sums of D*H*W outputs that differ by up to an ulp each, with random signs
    assert torch.allclose(t1, t0, rtol=5e-3, atol=8 * ulp * (D * H * W) ** 0.5), float((t1 - t0).abs().max())
    # dgrad into the two sources
    dy = rnd(N, D, H, W, cout, seed=35).to(dtype).to(DEV)
    wt = ops3d._w_taps(w, dtype, 1)
    wt_all = torch.stack([t.view(cin, 9, cout) for t in wt[::-1]], dim=2).contiguous()
    a0 = torch.zeros_like(x0)
    a1 = None if x1 is None else torch.zeros_like(x1)
    L.conv3d_dgrad(dy, wt, a0, a1, ksize=3)
    b0 = torch.full_like(x0, float("nan"))
    b1 = None if x1 is None else torch.full_like(x1, float("nan"))
    assert L.conv3d_dgrad_fused(dy, wt_all, b0, b1, ksize=3)
    torch.cuda.synchronize()
    for a, b in ((a0, b0), (a1, b1)):
        if a is not None:
            d = (b.float() - a.float()).abs()
            assert bool((d <= 4 * ulp * (a.float().abs() + 1.0)).all()), float(d.max())


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_one_launch_operands_from_the_parameter_equal_the_stacked_per_tap_packs(dtype):
    """ops3d._w_all (one cast-and-permute copy of the Conv3d parameter) against the construction it replaces: fi_pack_weights of
    every depth tap (mode 0 forward, mode 1 dgrad) stacked over the taps -- in reverse for the dgrad operand.  Bit for bit."""
    from fedicra_amd import _lib as L
    from fedicra_amd import ops3d
    torch.manual_seed(4)
    for cout, cin in ((16, 48), (32, 16), (24, 8)):
        w = torch.nn.Parameter(torch.randn(cout, cin, 3, 3, 3, device="cuda"))
        packs = {0: [], 1: []}
        for mode in (0, 1):
            for t in range(3):
                wk = w.detach()[:, :, t].permute(0, 2, 3, 1).contiguous().float()
                dst = torch.empty(cout * 9 * cin, dtype=dtype, device="cuda")
                L.pack_weights(wk, dst, cout, 9, cin, mode)
                packs[mode].append(dst)
        want0 = torch.stack([t.view(cout, 9, cin) for t in packs[0]], dim=2).contiguous()
        want1 = torch.stack([t.view(cin, 9, cout) for t in packs[1][::-1]], dim=2).contiguous()
        assert torch.equal(ops3d._w_all(w, dtype, 0), want0)
        assert torch.equal(ops3d._w_all(w, dtype, 1), want1)


def test_flat_store_model_takes_its_3d_gradients_in_the_sinks():
    """A unet_3D on the flat store: after backward every parameter's .grad IS its slice of the flat gradient buffer (no foreign
    tensor for the optimizer to gather), and the gradients equal those of a second backward accumulated on top (x 2)."""
    from fedicra_amd import ops
    from fedicra_amd.networks.net_factory_3d import net_factory_3d
    from fedicra_amd.networks.unet import set_compute_dtype
    torch.manual_seed(1)
    net = net_factory_3d("unet_3D", 1, 2).cuda().train()
    set_compute_dtype(net, "bf16")
    for m in net.modules():                                               # (every forward draws fresh dropout masks: off for the x 2 check)
        if isinstance(m, (torch.nn.Dropout, torch.nn.Dropout3d)):
            m.p = 0.0
    x = torch.rand(1, 1, 32, 32, 32, device="cuda")
    y = (torch.rand(1, 32, 32, 32, device="cuda") > 0.5).to(torch.uint8)

    def backward():
        ops.begin_iteration(torch.device("cuda"))
        out = net(x).permute(0, 2, 3, 4, 1)
        N, D, H, W, C = out.shape
        ops.ce_loss(out.reshape(N * D, H, W, C), y.reshape(N * D, H, W), 255).backward()

    net.zero_grad()
    backward()
    torch.cuda.synchronize()
    sinks = 0
    for p in net.parameters():
        if p.grad is not None and getattr(p, "_fi_gview", None) is not None:
            assert p.grad.data_ptr() == p._fi_gview.data_ptr()
            sinks += 1
    assert sinks > 30
    g1 = net.flat_grads.clone()
    backward()                                                            # accumulates
    torch.cuda.synchronize()
    g2 = net.flat_grads.clone()
    assert float(g1.abs().max()) > 0
    assert torch.allclose(g2, 2 * g1, rtol=2e-2, atol=1e-3 * float(g1.abs().max()))     # (atomic statistics sums: not bit-identical runs)
