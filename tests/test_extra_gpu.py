"""The reference's defined-but-never-run module surface against the CPU ops it names:
ConvTranspose2d / ConvTranspose3d(kernel 2, stride 2), GroupNorm (+ ReLU), UpBlock(bilinear=False)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda"
DTYPES = [torch.float32, torch.bfloat16, torch.float16]


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.rand(*shape, generator=g) * 2 - 1) * scale


def close(got, ref, dtype, what, k=1.0):
    s = float(ref.abs().max()) + 1e-6
    tol = (3e-5 if dtype == torch.float32 else 2e-2) * s * k
    err = float((got.float().cpu() - ref).abs().max())
    assert err <= tol, f"{what}: max err {err:.3e} (tol {tol:.1e})"


def cl(t, dtype):                       # NCHW / NCDHW -> dense channel-last in `dtype` on the device
    perm = (0, 2, 3, 1) if t.dim() == 4 else (0, 2, 3, 4, 1)
    return t.permute(*perm).contiguous().to(dtype).to(DEV)


def cf(t):                              # channel-last -> channel-first fp32 on the host
    perm = (0, 3, 1, 2) if t.dim() == 4 else (0, 4, 1, 2, 3)
    return t.float().cpu().permute(*perm)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("shape", [(2, 16, 8, 5, 7), (1, 32, 16, 4, 4), (2, 16, 8, 3, 4, 5), (1, 8, 16, 2, 3, 2)])
def test_conv_transpose_k2s2_against_torch(shape, dtype):
    from fedicra_amd import extra_ops
    three_d = len(shape) == 6
    N, cin, cout = shape[:3]
    sp = shape[3:]
    mod = (torch.nn.ConvTranspose3d if three_d else torch.nn.ConvTranspose2d)(cin, cout, kernel_size=2, stride=2)
    with torch.no_grad():
        mod.weight.copy_(rnd(*mod.weight.shape, seed=1, scale=0.3).to(dtype).float())
        mod.bias.copy_(rnd(cout, seed=2, scale=0.2))
    x = rnd(N, cin, *sp, seed=3).to(dtype).float().requires_grad_(True)
    y = mod(x)
    g = rnd(*y.shape, seed=4).to(dtype).float()
    y.backward(g)
    ref = (y.detach(), x.grad.clone(), mod.weight.grad.clone(), mod.bias.grad.clone())
    dmod = (torch.nn.ConvTranspose3d if three_d else torch.nn.ConvTranspose2d)(cin, cout, kernel_size=2, stride=2).to(DEV)
    with torch.no_grad():
        dmod.weight.copy_(mod.weight)
        dmod.bias.copy_(mod.bias)
    xd = cl(x.detach(), dtype).requires_grad_(True)
    yd = extra_ops.conv_transpose2x(xd, dmod)
    assert tuple(yd.shape) == (N,) + tuple(2 * s for s in sp) + (cout,)
    yd.backward(cl(g, dtype))
    close(cf(yd.detach()), ref[0], dtype, "fwd")
    close(cf(xd.grad), ref[1], dtype, "dx", 2.0)
    close(dmod.weight.grad, ref[2], dtype, "dW", 4.0)
    close(dmod.bias.grad, ref[3], dtype, "db", 4.0)
    with pytest.raises(NotImplementedError):
        extra_ops.conv_transpose2x(xd, torch.nn.ConvTranspose2d(cin, cout, 3, stride=2).to(DEV))


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("case", [(2, 32, 16, (6, 5), True), (1, 16, 16, (4, 4, 3), False), (3, 64, 16, (7,), True),
                                  (2, 16, 4, (5, 5), False)])
def test_group_norm_against_torch(case, dtype):
    from fedicra_amd import extra_ops
    N, C, G, sp, relu = case
    gn = torch.nn.GroupNorm(G, C)
    with torch.no_grad():
        gn.weight.copy_(rnd(C, seed=5) + 1.5)
        gn.bias.copy_(rnd(C, seed=6, scale=0.5))
    x = (rnd(N, C, *sp, seed=7, scale=2.0) + 0.3).to(dtype).float().requires_grad_(True)
    z = gn(x)
    if relu:
        z = F.relu(z)
    g = rnd(*z.shape, seed=8).to(dtype).float()
    z.backward(g)
    dgn = torch.nn.GroupNorm(G, C).to(DEV)
    with torch.no_grad():
        dgn.weight.copy_(gn.weight)
        dgn.bias.copy_(gn.bias)
    perm = (0,) + tuple(range(2, 2 + len(sp))) + (1,)
    inv = (0, len(sp) + 1) + tuple(range(1, len(sp) + 1))
    xd = x.detach().permute(*perm).contiguous().to(dtype).to(DEV).requires_grad_(True)
    zd = extra_ops.group_norm(xd, dgn, relu=relu)
    zd.backward(g.permute(*perm).contiguous().to(dtype).to(DEV))
    close(zd.detach().float().cpu().permute(*inv), z.detach(), dtype, "fwd")
    close(xd.grad.float().cpu().permute(*inv), x.grad, dtype, "dx", 4.0)
    close(dgn.weight.grad, gn.weight.grad, dtype, "dgamma", 4.0)
    close(dgn.bias.grad, gn.bias.grad, dtype, "dbeta", 4.0)


def test_upblock_transposed_conv_branch():
    """UpBlock(bilinear=False) (unet.py:60-62): state_dict keys of the reference's module, and the forward / backward equal
    ConvBlock(cat([x2, ConvTranspose2d(x1)])) assembled from the same parameters with torch's transposed convolution."""
    from fedicra_amd.networks.unet import ConvBlock, UpBlock
    torch.manual_seed(5)
    up = UpBlock(32, 16, 16, 0.0, bilinear=False).to(DEV).train()
    assert list(up.state_dict())[:2] == ["up.weight", "up.bias"] and "conv1x1.weight" not in up.state_dict()
    x1 = rnd(2, 32, 6, 5, seed=1).to(DEV).requires_grad_(True)
    x2 = rnd(2, 16, 12, 10, seed=2).to(DEV).requires_grad_(True)
    out = up(x1, x2)
    out.square().mean().backward()
    got = (out.detach().cpu(), x1.grad.cpu().clone(), up.up.weight.grad.cpu().clone())
    # reference composition: torch's ConvTranspose2d on the host, then this package's (already validated) ConvBlock
    ref_t = torch.nn.ConvTranspose2d(32, 16, 2, stride=2)
    ref_t.load_state_dict({"weight": up.up.weight.detach().cpu(), "bias": up.up.bias.detach().cpu()})
    blk = ConvBlock(32, 16, 0.0).to(DEV).train()
    blk.load_state_dict(up.conv.state_dict())
    h1 = x1.detach().cpu().requires_grad_(True)
    mid = ref_t(h1)
    cat = torch.cat([x2.detach().cpu(), mid], dim=1).to(DEV)
    cat = cat.detach().requires_grad_(True)
    o2 = blk(cat)
    o2.square().mean().backward()
    mid.backward(cat.grad[:, 16:].cpu())
    assert (got[0] - o2.detach().cpu()).abs().max().item() < 1e-4
    assert (got[1] - h1.grad).abs().max().item() < 1e-4 * max(1.0, h1.grad.abs().max().item())
    assert (got[2] - ref_t.weight.grad).abs().max().item() < 1e-4 * max(1.0, ref_t.weight.grad.abs().max().item())


@pytest.mark.parametrize("nz", ["none", "batchnorm", "groupnorm", "instancenorm"])
def test_vnet_matches_reference_golden(golden, nz):
    """VNet (a18) against the reference's own module (golden g12): same constructor + torch seed -> same initial state, then
    train-mode logits, loss, every parameter gradient (checksums) and BatchNorm3d's running statistics."""
    from fedicra_amd.networks.unet import set_compute_dtype
    from fedicra_amd.networks.vnet import VNet
    from helpers import assert_ck
    g = golden("g12_vnet.npz")
    torch.manual_seed(21)
    m = VNet(n_channels=1, n_classes=2, normalization=nz, has_dropout=False).cuda().train()
    set_compute_dtype(m, "fp32")
    x = torch.from_numpy(g[f"{nz}/x"]).to(DEV)
    y = torch.from_numpy(g[f"{nz}/y"]).long().to(DEV)
    out = m(x)
    ref = torch.from_numpy(g[f"{nz}/logits"])
    err = (out.detach().cpu() - ref).abs().max().item()
    assert err < 3e-4 * max(1.0, ref.abs().max().item()), f"logits err {err:.3e}"
    loss = F.cross_entropy(out.float(), y)
    assert abs(loss.item() - float(g[f"{nz}/loss"])) < 2e-5
    loss.backward()
    worst = 0.0
    for k, p in m.named_parameters():
        assert p.grad is not None, k
        ck = g[f"{nz}/grad/{k}"]
        if nz != "none" and k.endswith(".bias") and (".conv.0." in k or ".conv.3." in k or ".conv.6." in k) and "out_conv" not in k \
                and nz in ("batchnorm", "groupnorm", "instancenorm") and p.dim() == 1 and k.split(".")[-2] in ("0", "3", "6"):
            continue                    # conv bias in front of a normalisation: true gradient 0, round-off on both sides
        # checksums of whole gradient tensors; ReLU'(0) round-off flips behind a normalisation move them by ~1e-3
        # (DESIGN.md "parity bar"), logits and loss above are the tight checks
        assert_ck(p.grad.double().cpu(), ck, rtol=1e-2, atol=2e-6, what=f"{nz} grad {k}")
    if nz == "batchnorm":
        bn = m.block_one.conv[1]
        assert (bn.running_mean.cpu() - torch.from_numpy(g[f"{nz}/running_mean0"])).abs().max().item() < 1e-5
        assert (bn.running_var.cpu() - torch.from_numpy(g[f"{nz}/running_var0"])).abs().max().item() < 1e-4
        assert int(bn.num_batches_tracked) == 1


def test_vnet_factory_dropout_and_bf16():
    from fedicra_amd import ops
    from fedicra_amd.networks.net_factory_3d import net_factory_3d
    from fedicra_amd.networks.unet import set_compute_dtype
    torch.manual_seed(1)
    m = net_factory_3d("vnet", 1, 2).train()
    assert m.has_dropout and isinstance(m.block_one.conv[1], torch.nn.BatchNorm3d)
    set_compute_dtype(m, "bf16")
    x = torch.rand(2, 1, 16, 16, 16, device=DEV)
    ops.manual_seed(2)
    a, b = m(x), m(x)
    assert a.shape == (2, 2, 16, 16, 16) and a.dtype == torch.float32 and not torch.equal(a, b)     # Dropout3d(0.5) draws
    assert torch.equal(m(x, turnoff_drop=True).isfinite().all(), torch.tensor(True, device=DEV))
    a.square().mean().backward()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in m.parameters())


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("kind", ["batchnorm", "batchnorm_eval", "none", "instancenorm"])
def test_norm_relu_tail_against_torch(kind, dtype):
    """The `normalization` + ReLU tail of VNet's blocks on an NDHWC volume: BatchNorm3d (batch statistics from
    fi_channel_stats, running statistics updated; eval mode), no normalisation, InstanceNorm3d."""
    from fedicra_amd import extra_ops
    N, C, sp = 2, 32, (3, 4, 5)
    x = (rnd(N, C, *sp, seed=1, scale=2.0) + 0.3).to(dtype).float().requires_grad_(True)
    mod = {"batchnorm": torch.nn.BatchNorm3d(C), "batchnorm_eval": torch.nn.BatchNorm3d(C), "none": None,
           "instancenorm": torch.nn.InstanceNorm3d(C)}[kind]
    if kind.startswith("batchnorm"):
        with torch.no_grad():
            mod.weight.copy_(rnd(C, seed=2) + 1.5)
            mod.bias.copy_(rnd(C, seed=3))
            mod.running_mean.copy_(rnd(C, seed=4))
            mod.running_var.copy_(rnd(C, seed=5) + 1.5)
        mod.train(kind == "batchnorm")
    import copy
    dmod = copy.deepcopy(mod).to(DEV) if mod is not None else None
    z = F.relu(mod(x) if mod is not None else x)
    g = rnd(*z.shape, seed=6).to(dtype).float()
    z.backward(g)
    xd = x.detach().permute(0, 2, 3, 4, 1).contiguous().to(dtype).to(DEV).requires_grad_(True)
    zd = extra_ops.norm_relu(xd, dmod)
    zd.backward(g.permute(0, 2, 3, 4, 1).contiguous().to(dtype).to(DEV))
    close(zd.detach().float().cpu().permute(0, 4, 1, 2, 3), z.detach(), dtype, "fwd")
    close(xd.grad.float().cpu().permute(0, 4, 1, 2, 3), x.grad, dtype, "dx", 4.0)
    if kind == "batchnorm":
        close(dmod.weight.grad, mod.weight.grad, dtype, "dgamma", 4.0)
        close(dmod.bias.grad, mod.bias.grad, dtype, "dbeta", 4.0)
        assert torch.allclose(dmod.running_mean.cpu(), mod.running_mean, atol=1e-2 if dtype != torch.float32 else 1e-5)
        assert torch.allclose(dmod.running_var.cpu(), mod.running_var, atol=2e-2 if dtype != torch.float32 else 1e-4)
        assert int(dmod.num_batches_tracked) == 1
