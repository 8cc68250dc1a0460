"""Parity at BASELINE.json's full sizes (configs[1]: U-Net(1,2), 12x1x256x256; configs[2]: unet_lc at 512^2).

Small-shape tests elsewhere compare element by element with the oracle; here every kernel runs at the shapes the bench
runs it at and is checked (1) directly against the CPU op where the CPU finishes in well under a second, and
(2) through size-independent properties: the adjoint identities that tie forward, dgrad and wgrad together
(<conv(x), dy> = <x, dgrad(dy)> = <w, wgrad(x, dy)>), the two zero-sums of a BatchNorm backward, conservation of the
gradient through max-pool, the adjoint of the bilinear up-sampling, and the whole training step against the oracle."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda"

# (H, c0, c1, cout, k): every conv of UNet(1,2) at 256^2, batch 12 (tools/kbench.py LAYERS)
LAYERS = [(256, 1, 0, 16, 3), (256, 16, 0, 16, 3), (128, 16, 0, 32, 3), (128, 32, 0, 32, 3), (64, 32, 0, 64, 3),
          (64, 64, 0, 64, 3), (32, 64, 0, 128, 3), (32, 128, 0, 128, 3), (16, 128, 0, 256, 3), (16, 256, 0, 256, 3),
          (16, 256, 0, 128, 1), (32, 128, 128, 128, 3), (32, 128, 0, 64, 1), (64, 64, 64, 64, 3), (64, 64, 0, 32, 1),
          (128, 32, 32, 32, 3), (128, 32, 0, 16, 1), (256, 16, 16, 16, 3), (256, 16, 0, 2, 3)]
N = 12


def L():
    from fedicra_amd import _lib
    return _lib


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.rand(*shape, generator=g) * 2 - 1) * scale


def dot(a, b):
    return float((a.double() * b.double()).sum())


def _run_layer(layer, dtype):
    """-> dict of device results and CPU inputs for one full-size layer."""
    H, c0, c1, cout, k = layer
    cin = c0 + c1
    x = rnd(N, H, H, cin, seed=1).to(dtype)
    w = rnd(cout, k, k, cin, seed=2, scale=1.0 / np.sqrt(cin * k * k)).to(dtype)        # [Cout][kh][kw][Cin]
    b = rnd(cout, seed=3, scale=0.5)
    dy = rnd(N, H, H, cout, seed=4).to(dtype)
    xd, wd, dyd = x.to(DEV), w.to(DEV), dy.to(DEV)
    x0 = xd[..., :c0].contiguous()
    x1 = xd[..., c0:].contiguous() if c1 else None
    y = torch.empty(N, H, H, cout, dtype=dtype, device=DEV)
    L().conv2d_fwd(x0, x1, wd, b.to(DEV), y, None, None, ksize=k)
    wt = torch.empty(cin * k * k * cout, dtype=dtype, device=DEV)
    L().pack_weights(wd.float(), wt, cout, k * k, cin, 1)
    d0 = torch.empty(N, H, H, c0, dtype=dtype, device=DEV)
    d1 = torch.empty(N, H, H, c1, dtype=dtype, device=DEV) if c1 else None
    L().conv2d_fwd(dyd, None, wt, None, d0, d1, None, ksize=k)
    dx = d0 if d1 is None else torch.cat([d0, d1], dim=3)
    dw = torch.zeros(cout, k, k, cin, dtype=torch.float32, device=DEV)
    db = torch.zeros(cout, dtype=torch.float32, device=DEV)
    L().conv2d_wgrad(x0, x1, dyd, dw, db, ksize=k)
    torch.cuda.synchronize()
    return dict(x=x, w=w, b=b, dy=dy, y=y, dx=dx, dw=dw, db=db, xd=xd, wd=wd, dyd=dyd)


@pytest.mark.parametrize("layer", LAYERS, ids=lambda l: "{}x{}_{}+{}to{}_k{}".format(l[0], l[0], *l[1:]))
def test_fp32_layer_at_bench_shape_matches_cpu_conv_and_adjoint_identities(layer):
    H, c0, c1, cout, k = layer
    r = _run_layer(layer, torch.float32)
    x = r["x"].permute(0, 3, 1, 2).requires_grad_(True)
    w = r["w"].permute(0, 3, 1, 2).requires_grad_(True)
    b = r["b"].clone().requires_grad_(True)
    ref = F.conv2d(x, w, b, padding=k // 2)
    ref.backward(r["dy"].permute(0, 3, 1, 2))
    for name, got, want in (("fwd", r["y"].permute(0, 3, 1, 2), ref.detach()), ("dgrad", r["dx"].permute(0, 3, 1, 2), x.grad),
                            ("wgrad", r["dw"].permute(0, 3, 1, 2), w.grad), ("dbias", r["db"], b.grad)):
        s = float(want.abs().max())
        err = float((got.cpu() - want).abs().max())
        assert err < 1e-4 * s + 1e-6, f"{name}: max err {err:.3e} of {s:.3e}"
    # adjoint identities, fp64 accumulation on the device (bias removed from the forward side)
    y0 = r["y"].double() - r["b"].to(DEV).double()
    a = dot(y0, r["dyd"])
    assert abs(a - dot(r["xd"], r["dx"])) < 1e-5 * abs(a) + 1e-3
    assert abs(a - dot(r["wd"], r["dw"])) < 1e-5 * abs(a) + 1e-3
    assert abs(float(r["db"].double().sum()) - float(r["dyd"].double().sum())) < 1e-5 * float(r["dyd"].abs().double().sum())


@pytest.mark.parametrize("layer", LAYERS, ids=lambda l: "{}x{}_{}+{}to{}_k{}".format(l[0], l[0], *l[1:]))
def test_bf16_layer_at_bench_shape_adjoint_identities_and_fp32_agreement(layer):
    """bf16 storage, fp32 accumulation: the three kernels see the same bf16 operands, so the identities hold up to the
    rounding of the stored outputs (y, dx: one bf16 rounding per element; dw stays fp32)."""
    H, c0, c1, cout, k = layer
    r = _run_layer(layer, torch.bfloat16)
    y0 = r["y"].double() - r["b"].to(DEV).double()
    a = dot(y0, r["dyd"])
    exact = dot(r["wd"], r["dw"])                      # fp32 accumulation of exact bf16 products: the reference value
    n_out = r["y"].numel()
    # a sum of n independently rounded terms: error ~ 2^-9 * rms(term) * sqrt(n); allow 6 sigma
    sig_y = 2.0 ** -9 * float((y0 * r["dyd"].double()).pow(2).sum().sqrt())
    sig_x = 2.0 ** -9 * float((r["xd"].double() * r["dx"].double()).pow(2).sum().sqrt())
    assert abs(a - exact) < 6 * sig_y + 1e-6 * abs(exact) + 2.0 ** -9 * abs(float(r["b"].abs().max())) * np.sqrt(n_out)
    assert abs(dot(r["xd"], r["dx"]) - exact) < 6 * sig_x + 1e-6 * abs(exact)
    # and element-wise against the same layer computed in fp32 from the same (bf16-valued) operands
    f = {kk: (v.float() if torch.is_tensor(v) else v) for kk, v in r.items()}
    x0 = f["xd"][..., :c0].contiguous()
    x1 = f["xd"][..., c0:].contiguous() if c1 else None
    y32 = torch.empty(N, H, H, cout, dtype=torch.float32, device=DEV)
    L().conv2d_fwd(x0, x1, f["wd"], r["b"].to(DEV), y32, None, None, ksize=k)
    s = float(y32.abs().max())
    assert float((r["y"].float() - y32).abs().max()) < 2.0 ** -8 * s + 1e-6
    dw32 = torch.zeros_like(r["dw"])
    L().conv2d_wgrad(x0, x1, f["dyd"], dw32, None, ksize=k)
    sw = float(dw32.abs().max())
    assert float((r["dw"] - dw32).abs().max()) < 2e-4 * sw + 1e-5, "bf16 and fp32 wgrad accumulate the same exact products"


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("shape", [(256, 16), (128, 32), (16, 256)])
def test_batchnorm_properties_at_bench_shape(shape, dtype):
    """Train-mode BN + LeakyReLU at 12 x H x H x C: normalised output has the affine's mean / variance; the backward's
    input gradient sums to zero and is orthogonal to the normalised activations, per channel."""
    H, C = shape
    lib = L()
    y = (rnd(N, H, H, C, seed=5, scale=2.0) + 0.3).to(dtype).to(DEV)
    yq = y.double()
    cnt = float(N * H * H)
    stats = torch.zeros(lib.STATS_SLOTS, C, 2, dtype=torch.float64, device=DEV)
    stats[0] = torch.stack([yq.sum((0, 1, 2)), (yq * yq).sum((0, 1, 2))], 1)
    g = (rnd(C, seed=6) + 1.5).to(DEV)
    be = rnd(C, seed=7).to(DEV)
    rm, rv = torch.zeros(C, device=DEV), torch.ones(C, device=DEV)
    nbt = torch.zeros(1, dtype=torch.int64, device=DEV)
    coef = torch.empty(4, C, device=DEV)
    lib.bn_finalize(stats, cnt, g, be, rm, rv, nbt, 0.1, 1e-5, True, coef[0], coef[1], coef[2], coef[3])
    z = torch.empty_like(y)
    lib.bn_act_fwd(y, coef[0], coef[1], z, 1.0, None)                 # slope 1: the affine-normalised tensor itself
    zq = z.double()
    tol_ = 1e-4 if dtype == torch.float32 else 1e-2
    assert float((zq.mean((0, 1, 2)) - be.double()).abs().max()) < tol_
    assert float((zq.var((0, 1, 2), unbiased=False).sqrt() - g.double().abs()).abs().max()) < tol_
    mean = yq.mean((0, 1, 2))
    assert torch.allclose(rm.double(), 0.1 * mean, rtol=1e-4, atol=1e-6)
    assert torch.allclose(rv.double(), 0.9 + 0.1 * yq.var((0, 1, 2), unbiased=True), rtol=1e-4)
    dz = rnd(N, H, H, C, seed=8).to(dtype).to(DEV)
    sums = torch.zeros(lib.STATS_SLOTS * 2 * C, dtype=torch.float64, device=DEV)
    lib.bn_act_bwd_reduce(dz, y, coef[0], coef[1], coef[2], coef[3], sums, 0.01, None)
    dy = torch.empty_like(y)
    dg, db = torch.zeros(C, device=DEV), torch.zeros(C, device=DEV)
    lib.bn_act_bwd_apply(dz, y, coef[0], coef[1], coef[2], coef[3], sums, True, dy, dg, db, 0.01, None)
    xhat = (yq - mean) / (yq.var((0, 1, 2), unbiased=False) + 1e-5).sqrt()
    scale = float(dy.double().abs().sum((0, 1, 2)).max())
    rel = 1e-5 if dtype == torch.float32 else 3e-3
    assert float(dy.double().sum((0, 1, 2)).abs().max()) < rel * scale
    assert float((dy.double() * xhat).sum((0, 1, 2)).abs().max()) < rel * scale * 2


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_pool_and_upsample_properties_at_bench_shape(dtype):
    lib = L()
    for H, C in [(256, 16), (64, 64)]:
        x = rnd(N, H, H, C, seed=9).to(dtype).to(DEV)
        y = torch.empty(N, H // 2, H // 2, C, dtype=dtype, device=DEV)
        lib.maxpool2_fwd(x, y)
        assert torch.equal(y, x.view(N, H // 2, 2, H // 2, 2, C).amax((2, 4)))
        dy = rnd(N, H // 2, H // 2, C, seed=10).to(dtype).to(DEV)
        dx = torch.empty_like(x)
        lib.maxpool2_bwd(x, dy, dx)
        assert torch.equal(dx.view(N, H // 2, 2, H // 2, 2, C).sum((2, 4)), dy)       # one winner per window, exact
        assert dot(dx, x) == pytest.approx(dot(dy, y), rel=1e-12)
    for h, C in [(128, 16), (16, 128)]:
        u = rnd(N, h, h, C, seed=11).to(dtype).to(DEV)
        up = torch.empty(N, 2 * h, 2 * h, C, dtype=dtype, device=DEV)
        lib.upsample2x_fwd(u, up)
        g = rnd(N, 2 * h, 2 * h, C, seed=12).to(dtype).to(DEV)
        du = torch.empty_like(u)
        lib.upsample2x_bwd(g, du)
        a, b = dot(up, g), dot(u, du)
        rel = 1e-5 if dtype == torch.float32 else 2e-3
        assert abs(a - b) < rel * max(abs(a), float((up.double() * g.double()).abs().sum()) * 1e-2)
        const = torch.full_like(u, 0.75)
        lib.upsample2x_fwd(const, up)
        assert float((up.float() - 0.75).abs().max()) < (1e-6 if dtype == torch.float32 else 4e-3)  # partition of unity


class _MaskDropout(torch.nn.Module):
    """nn.Dropout whose keep-mask is drawn in fp32 whatever the activation dtype: the fp64 run then consumes the generator
    exactly like the fp32 oracle and like the mask provider handed to the HIP path."""

    def __init__(self, p):
        super().__init__()
        self.p = p

    def forward(self, x):
        if not self.training or self.p == 0:
            return x
        return x * torch.empty(x.shape).bernoulli_(1 - self.p).to(x.dtype) / (1 - self.p)


def _swap_dropout(mod):
    for name, child in mod.named_children():
        if isinstance(child, torch.nn.Dropout):
            setattr(mod, name, _MaskDropout(child.p))
        else:
            _swap_dropout(child)
    return mod


def test_config2_training_step_at_full_size_matches_oracle():
    """BASELINE.json configs[1] exactly: one forward + backward of UNet(1,2) on 12x1x256x256 with the reference's dropout
    masks.  At 9.4 M activations per layer some BatchNorm outputs always lie within round-off of 0, where LeakyReLU's
    derivative jumps (DESIGN.md "parity bar": two correct fp32 implementations then differ by ~1e-2 of a gradient's
    scale, the reference against itself included).  So the yardstick is an fp64 run of the oracle: per parameter, the
    HIP gradient must be as close to it as the fp32 CPU oracle's own gradient is (x3), and the two fp32 losses agree."""
    from fedicra_amd import ops
    from fedicra_amd.networks.unet import UNet, set_compute_dtype
    from oracle.losses_ref import pce_loss
    from oracle.unet_ref import RefUNet, seeded_state
    from helpers import loader
    b = loader(1, 12, 256, cid=0)[0]
    x, label = b["image"].unsqueeze(1), b["label"]

    def mkref(dtype):
        r = RefUNet(1, 2)
        seeded_state(r, 2022)
        return _swap_dropout(r).to(dtype).train()

    grads, losses = {}, {}
    for dtype in (torch.float32, torch.float64):
        r = mkref(dtype)
        torch.manual_seed(3)
        loss = pce_loss(r(x.to(dtype))[0], label, 2)
        loss.backward()
        grads[dtype] = {n: p.grad.double() for n, p in r.named_parameters()}
        losses[dtype] = float(loss.detach())
    m = UNet(1, 2)
    seeded_state(m, 2022)
    m = m.cuda().train()
    set_compute_dtype(m, "fp32")
    ops.set_dropout_mask_provider(lambda shape, p: torch.empty(shape).bernoulli_(1 - p))
    try:
        torch.manual_seed(3)
        loss = ops.ce_loss(m(x.to(DEV))[0].permute(0, 2, 3, 1), label.to(DEV), 2)
        loss.backward()
    finally:
        ops.set_dropout_mask_provider(None)
    assert abs(float(loss.detach()) - losses[torch.float64]) < 3 * abs(losses[torch.float32] - losses[torch.float64]) + 2e-6
    worst_hip = worst_cpu = 0.0
    for n, p in m.named_parameters():
        if n.endswith("conv_conv.0.bias") or n.endswith("conv_conv.4.bias"):
            continue                                  # conv bias before BN: true gradient 0, only round-off on every side
        g64 = grads[torch.float64][n]
        s = max(float(g64.abs().max()), 1e-9)
        e_cpu = float((grads[torch.float32][n] - g64).abs().max()) / s
        e_hip = float((p.grad.double().cpu() - g64).abs().max()) / s
        worst_hip, worst_cpu = max(worst_hip, e_hip), max(worst_cpu, e_cpu)
        assert e_hip < 3 * e_cpu + 2e-4, f"{n}: HIP vs fp64 {e_hip:.3e}, CPU fp32 vs fp64 {e_cpu:.3e}"
    print(f"configs[1] full size, worst rel. gradient error vs fp64: HIP {worst_hip:.3e}, CPU fp32 oracle {worst_cpu:.3e}")


def test_config3_unet_lc_512_forward_matches_oracle():
    """configs[2] shape (unet_lc, 512^2; batch 2 keeps the CPU side to seconds): eval-mode outputs vs the oracle."""
    from fedicra_amd.networks.unet import UNet_LC, set_compute_dtype
    from oracle.unet_ref import RefUNetLC, seeded_state
    from helpers import loader
    b = loader(1, 2, 512, cid=3)[0]
    x = b["image"].unsqueeze(1)
    ref = RefUNetLC(1, 2, 1, 8, 8, 3)
    m = UNet_LC(1, 2, 1, 8, 8, 3)
    for mod in (ref, m):
        extra = {f"encoder.pcs_list.{i}.{k}": v for i, p in enumerate(mod.encoder.pcs_list) for k, v in p.state_dict().items()}
        seeded_state(mod, 2022, extra=extra)
    ref.eval()
    m = m.cuda().eval()
    set_compute_dtype(m, "fp32")
    with torch.no_grad():
        want = ref(x, None)
        got = m(x.to(DEV), None)
    assert (got[0].cpu() - want[0]).abs().max().item() < 2e-4 * max(1.0, want[0].abs().max().item())
    assert (got[6][-1].cpu() - want[6][-1]).abs().max().item() < 1e-5


# ----------------------------------------------------------------------------------------------- configs[3]: 3D U-Net 128^3, bf16
# (S, c0, c1, cout): 3x3x3 convs of unet_3D (feature_scale 4) at the sizes a 2 x 1 x 128^3 patch batch runs them
C4_LAYERS = [(128, 16, 0, 16), (64, 32, 0, 32), (32, 64, 0, 64), (32, 64, 32, 32), (64, 32, 16, 16)]


@pytest.mark.parametrize("dtype", ["bf16", "fp16"])
@pytest.mark.parametrize("layer", C4_LAYERS, ids=lambda l: "{}^3_{}+{}to{}".format(*l))
def test_config4_conv3d_instancenorm_at_full_size_bf16(layer, dtype):
    """configs[3] (4 clients, 3D U-Net, 2 x 128^3 patches, bf16) and configs[4]'s storage type (fp16) at the same layer sizes,
    through size-independent properties:
    (a) the adjoint identities tying the depth-sliced forward, dgrad and wgrad together -- <conv(x; w), g> = <x, dgrad(g)>
    = <w, wgrad(x, g)> (bias 0), all three products of 16-bit operands accumulated in fp32, compared in fp64 to bf16 output
    rounding (2^-8 per element, random signs: relative 4e-3 on sums of 1e6+ terms);  (b) the fused InstanceNorm3d + ReLU
    (per-sample fp64 statistics accumulated by the centre depth tap's epilogue) against fp32 InstanceNorm + ReLU of the HIP
    convolution's own output: element-wise to bf16 resolution, and the first two moments per (sample, channel) over the
    whole volume to 0.5 % / 1 %."""
    from fedicra_amd import ops3d
    S, c0, c1, cout = layer
    Nb, cin, dt = 2, c0 + c1, (torch.bfloat16 if dtype == "bf16" else torch.float16)
    conv = torch.nn.Conv3d(cin, cout, 3, 1, 1).to(DEV)
    with torch.no_grad():
        conv.weight.copy_((rnd(*conv.weight.shape, seed=21) * (1.7 / np.sqrt(27 * cin))).to(dt).float())
        conv.bias.zero_()
    x0 = rnd(Nb, S, S, S, c0, seed=22).to(dt).to(DEV).requires_grad_(True)
    x1 = rnd(Nb, S, S, S, c1, seed=23).to(dt).to(DEV).requires_grad_(True) if c1 else None
    g = rnd(Nb, S, S, S, cout, seed=24).to(dt).to(DEV)
    y = ops3d.conv3d(x0, x1, conv, norm=False)
    assert y.dtype == dt and y.shape == (Nb, S, S, S, cout)
    y.backward(g)
    torch.cuda.synchronize()
    lhs = dot(y.detach(), g)
    via_x = dot(x0.detach(), x0.grad) + (dot(x1.detach(), x1.grad) if c1 else 0.0)
    via_w = dot(conv.weight.detach(), conv.weight.grad)
    scale = max(abs(lhs), float(y.detach().float().norm()) * float(g.float().norm()) * 1e-3)
    assert abs(lhs - via_x) <= 4e-3 * scale, (lhs, via_x)
    assert abs(lhs - via_w) <= 4e-3 * scale, (lhs, via_w)
    # (b) fused InstanceNorm + ReLU of the same convolution
    with torch.no_grad():
        z = ops3d.conv3d(x0.detach(), None if x1 is None else x1.detach(), conv, norm=True)
        yf = y.detach().float()
        mean = yf.mean(dim=(1, 2, 3), keepdim=True)
        var = yf.var(dim=(1, 2, 3), unbiased=False, keepdim=True)
        ref = torch.relu((yf - mean) * torch.rsqrt(var + 1e-5))
        err = (z.float() - ref).abs()
        assert float(err.max()) <= 3e-2 * max(1.0, float(ref.max())), float(err.max())
        assert float(err.mean()) <= 2e-3, float(err.mean())
        assert float((z < 0).sum()) == 0
        # moments of the normalised volume, from the HIP output: E[relu(n)] and E[relu(n)^2] of a unit normal-ish field
        # are not fixed numbers, the reference's are: compare the two per (sample, channel)
        m1, m1r = z.float().mean(dim=(1, 2, 3)), ref.mean(dim=(1, 2, 3))
        m2, m2r = (z.float() ** 2).mean(dim=(1, 2, 3)), (ref ** 2).mean(dim=(1, 2, 3))
        assert torch.allclose(m1, m1r, rtol=5e-3, atol=2e-3) and torch.allclose(m2, m2r, rtol=1e-2, atol=2e-3)
