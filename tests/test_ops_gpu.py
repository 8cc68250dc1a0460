"""GPU parity of every C-ABI kernel against the same op evaluated by torch on the CPU (fp32).

fp32 mode (v_mfma_f32_16x16x4_f32, exact fp32 products) must agree to fp32 round-off; bf16 mode
is checked against the CPU op applied to bf16-rounded operands, with a bf16-sized tolerance.
All calls go through fedicra_amd._lib, i.e. the C ABI of libfedicra_hip.so.
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

DEV = "cuda"
DTYPES = [torch.float32, torch.bfloat16, torch.float16]


def L():
    from fedicra_amd import _lib
    return _lib


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.rand(*shape, generator=g) * 2 - 1) * scale


def q(t, dtype):
    """round through the storage dtype (CPU)"""
    return t.to(dtype).float()


def nhwc(t, dtype):
    return t.permute(0, 2, 3, 1).contiguous().to(dtype).to(DEV)


def krsc(w, dtype):
    return w.permute(0, 2, 3, 1).contiguous().to(dtype).to(DEV)


def tol(dtype, ref):
    s = float(ref.abs().max()) + 1e-6
    return (3e-5 * s, 1e-5) if dtype == torch.float32 else (2e-2 * s, 2e-2)


def close(got, ref, dtype, what=""):
    atol, rtol = tol(dtype, ref)
    got = got.float().cpu()
    err = (got - ref).abs().max().item()
    assert torch.allclose(got, ref, atol=atol, rtol=rtol), f"{what}: max err {err:.3e} (atol {atol:.1e})"


CONV_CASES = [
    # N, H, W, c0, c1, cout, k
    (2, 20, 37, 1, 0, 16, 3), (2, 20, 37, 3, 0, 16, 3), (2, 33, 16, 16, 0, 16, 3), (1, 64, 64, 16, 0, 32, 3),
    (2, 18, 18, 32, 0, 32, 3), (3, 8, 8, 64, 0, 64, 3), (2, 4, 4, 128, 0, 256, 3), (2, 16, 24, 32, 32, 32, 3),
    (2, 32, 32, 16, 16, 16, 3), (2, 9, 9, 256, 0, 128, 1), (2, 16, 16, 32, 0, 16, 1), (2, 40, 24, 16, 0, 2, 3),
    (2, 16, 16, 512, 0, 3, 1), (4, 1, 1, 8, 0, 256, 1), (4, 1, 1, 256, 256, 16, 1), (1, 64, 48, 64, 0, 512, 3),
    (12, 64, 64, 64, 64, 64, 3),
]


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("case", CONV_CASES)
def test_conv_fwd_stats(case, dtype):
    N, H, W, c0, c1, cout, k = case
    cin = c0 + c1
    x = q(rnd(N, cin, H, W, seed=1), dtype)
    w = q(rnd(cout, cin, k, k, seed=2, scale=1.0 / np.sqrt(cin * k * k)), dtype)
    b = rnd(cout, seed=3, scale=0.5)
    ref = F.conv2d(x, w, b, padding=k // 2)
    x0 = nhwc(x[:, :c0], dtype)
    x1 = nhwc(x[:, c0:], dtype) if c1 else None
    y = torch.empty(N, H, W, cout, dtype=dtype, device=DEV)
    stats = torch.zeros(L().STATS_SLOTS * cout * 2, dtype=torch.float64, device=DEV)
    L().conv2d_fwd(x0, x1, krsc(w, dtype), b.to(DEV), y, None, stats, ksize=k)
    torch.cuda.synchronize()
    close(y.permute(0, 3, 1, 2), ref, dtype, "conv fwd")
    yq = y.float().cpu().double()
    st = stats.cpu().view(L().STATS_SLOTS, cout, 2).sum(0)
    assert torch.allclose(st[:, 0], yq.sum((0, 1, 2)), rtol=1e-5, atol=1e-3)
    assert torch.allclose(st[:, 1], (yq * yq).sum((0, 1, 2)), rtol=1e-5, atol=1e-3)


def test_conv_fwd_f32_output_and_accumulate():
    N, H, W, cin, cout = 2, 24, 24, 16, 2
    x = q(rnd(N, cin, H, W, seed=1), torch.bfloat16)
    w = q(rnd(cout, cin, 3, 3, seed=2, scale=0.1), torch.bfloat16)
    b = rnd(cout, seed=3)
    ref = F.conv2d(x, w, b, padding=1)
    y = torch.empty(N, H, W, cout, dtype=torch.float32, device=DEV)
    L().conv2d_fwd(nhwc(x, torch.bfloat16), None, krsc(w, torch.bfloat16), b.to(DEV), y, None, None, ksize=3, y_f32=True)
    close(y.permute(0, 3, 1, 2), ref, torch.float32, "bf16 conv with fp32 logits")
    y2 = torch.full((N, H, W, cout), 1.5, dtype=torch.float32, device=DEV)
    xf = rnd(N, cin, H, W, seed=1)
    wf = rnd(cout, cin, 3, 3, seed=2, scale=0.1)
    L().conv2d_fwd(nhwc(xf, torch.float32), None, krsc(wf, torch.float32), None, y2, None, None, ksize=3, acc0=True)
    close(y2.permute(0, 3, 1, 2), F.conv2d(xf, wf, None, padding=1) + 1.5, torch.float32, "accumulate")


def test_tr16_semantics():
    """ds_read_b64_tr_b16 on gfx950: within each 16-lane group, lane i receives element (i & 3) of the 8 bytes
    fetched by lanes 4j + (i >> 2), j = 0..3.  The bf16 wgrad fragment loader is built on exactly this."""
    lib = L()
    inp = torch.arange(1024, dtype=torch.int16)
    g = torch.Generator().manual_seed(0)
    offs = (torch.randint(0, 255, (64,), generator=g) * 4).to(torch.int32)      # 8-byte aligned, arbitrary
    out = torch.empty(256, dtype=torch.int16, device=DEV)
    lib.probe_tr16(inp.to(DEV), offs.to(DEV), out)
    got = out.cpu().view(64, 4)
    exp = torch.empty(64, 4, dtype=torch.int16)
    for l in range(64):
        grp, i = l >> 4, l & 15
        for j in range(4):
            src_lane = grp * 16 + 4 * j + (i >> 2)
            exp[l, j] = inp[offs[src_lane] + (i & 3)]
    assert torch.equal(got, exp), f"transpose-read semantics differ:\n{got[:16]}\nexpected\n{exp[:16]}\noffs {offs[:16]}"


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("case", CONV_CASES)
def test_conv_dgrad_wgrad(case, dtype):
    N, H, W, c0, c1, cout, k = case
    if N * H * W * cout > 4_000_000:
        N = 2
    cin = c0 + c1
    x = q(rnd(N, cin, H, W, seed=4), dtype).requires_grad_(True)
    w = q(rnd(cout, cin, k, k, seed=5, scale=1.0 / np.sqrt(cin * k * k)), dtype).requires_grad_(True)
    b = rnd(cout, seed=6).requires_grad_(True)
    dy = q(rnd(N, cout, H, W, seed=7), dtype)
    F.conv2d(x, w, b, padding=k // 2).backward(dy)
    # dgrad: conv of dy with the flipped/transposed filter, two destinations for a concat input
    wt = torch.empty(cin * k * k * cout, dtype=dtype, device=DEV)
    L().pack_weights(krsc(w.detach(), torch.float32), wt, cout, k * k, cin, 1)
    d0 = torch.empty(N, H, W, c0, dtype=dtype, device=DEV)
    d1 = torch.empty(N, H, W, c1, dtype=dtype, device=DEV) if c1 else None
    dyd = nhwc(dy, dtype)
    L().conv2d_fwd(dyd, None, wt, None, d0, d1, None, ksize=k)
    got = d0 if d1 is None else torch.cat([d0, d1], dim=3)
    close(got.permute(0, 3, 1, 2), x.grad, dtype, "dgrad")
    dw = torch.zeros(cout, k, k, cin, dtype=torch.float32, device=DEV)
    db = torch.zeros(cout, dtype=torch.float32, device=DEV)
    xd = x.detach()
    L().conv2d_wgrad(nhwc(xd[:, :c0], dtype), nhwc(xd[:, c0:], dtype) if c1 else None, dyd, dw, db, ksize=k)
    torch.cuda.synchronize()
    at, rt = tol(dtype, w.grad)
    gw = dw.permute(0, 3, 1, 2).cpu()
    assert torch.allclose(gw, w.grad, atol=at * 3, rtol=rt), f"wgrad max err {(gw - w.grad).abs().max():.3e}"
    assert torch.allclose(db.cpu(), b.grad, atol=tol(dtype, b.grad)[0] * 3, rtol=rt), "dbias"


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("mode", ["train_drop", "train", "eval", "relu_chan"])
def test_bn_act_fwd_bwd(dtype, mode):
    N, C, H, W = 3, 32, 10, 14
    slope = 0.0 if mode == "relu_chan" else 0.01
    y = q(rnd(N, C, H, W, seed=1, scale=2.0) + 0.3, dtype).requires_grad_(True)
    bn = torch.nn.BatchNorm2d(C)
    with torch.no_grad():
        bn.weight.copy_(rnd(C, seed=2) + 1.5)
        bn.bias.copy_(rnd(C, seed=3))
        bn.running_mean.copy_(rnd(C, seed=4))
        bn.running_var.copy_(rnd(C, seed=5) + 1.5)
    training = mode != "eval"
    bn.train(training)
    p = {"train_drop": 0.3, "relu_chan": 0.25}.get(mode, 0.0)
    if mode == "relu_chan":
        keep = (torch.rand(N, C, 1, 1, generator=torch.Generator().manual_seed(9)) >= p).float()
    else:
        keep = (torch.rand(N, C, H, W, generator=torch.Generator().manual_seed(9)) >= p).float()
    rm0, rv0 = bn.running_mean.clone(), bn.running_var.clone()
    z = F.leaky_relu(bn(y), slope) if slope else F.relu(bn(y))
    if p:
        z = z * keep * (1.0 / (1.0 - p))
    dz = q(rnd(N, C, H, W, seed=6), dtype)
    z.backward(dz)
    # device
    lib = L()
    yd = nhwc(y.detach(), dtype)
    yq = yd.float().double()
    stats = torch.zeros(lib.STATS_SLOTS, C, 2, dtype=torch.float64, device=DEV)
    stats[3] = torch.stack([yq.sum((0, 1, 2)), (yq * yq).sum((0, 1, 2))], 1)      # any slot: finalize sums them
    g, be = bn.weight.detach().to(DEV), bn.bias.detach().to(DEV)
    rm, rv = rm0.to(DEV), rv0.to(DEV)
    nbt = torch.zeros(1, dtype=torch.int64, device=DEV)
    coef = torch.empty(4, C, device=DEV)
    lib.bn_finalize(stats if training else None, float(N * H * W), g, be, rm, rv, nbt, 0.1, 1e-5, training, coef[0],
                    coef[1], coef[2], coef[3])
    drop = None
    if p:
        if mode == "relu_chan":
            drop = (lib.DROP_MASK_CHAN, p, 0, keep.reshape(N, C).to(torch.uint8).to(DEV), None)
        else:
            drop = (lib.DROP_MASK_ELEM, p, 0, keep.permute(0, 2, 3, 1).contiguous().to(torch.uint8).to(DEV), None)
    zd = torch.empty_like(yd)
    lib.bn_act_fwd(yd, coef[0], coef[1], zd, slope, drop)
    close(zd.permute(0, 3, 1, 2), z.detach(), dtype, "bn_act fwd")
    if training:
        assert int(nbt) == 1
        assert torch.allclose(rm.cpu(), bn.running_mean, atol=1e-5) and torch.allclose(rv.cpu(), bn.running_var, atol=1e-4)
    dzd = nhwc(dz, dtype)
    sums = torch.zeros(lib.STATS_SLOTS * 2 * C, dtype=torch.float64, device=DEV)
    lib.bn_act_bwd_reduce(dzd, yd, coef[0], coef[1], coef[2], coef[3], sums, slope, drop)
    dyd = torch.empty_like(yd)
    dg, db = torch.zeros(C, device=DEV), torch.zeros(C, device=DEV)
    lib.bn_act_bwd_apply(dzd, yd, coef[0], coef[1], coef[2], coef[3], sums, training, dyd, dg, db, slope, drop)
    close(dyd.permute(0, 3, 1, 2), y.grad, dtype, "bn_act bwd dy")
    close(dg, bn.weight.grad, dtype, "dgamma")
    close(db, bn.bias.grad, dtype, "dbeta")


def test_dropout_rng_statistics_and_replay():
    lib = L()
    N, H, W, C = 4, 32, 32, 16
    y = torch.ones(N, H, W, C, device=DEV)
    one, zero = torch.ones(C, device=DEV), torch.zeros(C, device=DEV)
    off = torch.zeros(1, dtype=torch.int32, device=DEV)
    z1, z2, z3 = torch.empty_like(y), torch.empty_like(y), torch.empty_like(y)
    lib.bn_act_fwd(y, one, zero, z1, 1.0, (lib.DROP_RNG_ELEM, 0.3, 1234, None, off))
    lib.bn_act_fwd(y, one, zero, z2, 1.0, (lib.DROP_RNG_ELEM, 0.3, 1234, None, off))
    off += 1
    lib.bn_act_fwd(y, one, zero, z3, 1.0, (lib.DROP_RNG_ELEM, 0.3, 1234, None, off))
    assert torch.equal(z1, z2)                       # same (seed, offset) -> same mask (backward regenerates it)
    assert not torch.equal(z1, z3)                   # the device counter changes the mask under graph replay
    keep = (z1 != 0).float().mean().item()
    assert abs(keep - 0.7) < 0.01
    assert abs(z1.max().item() - 1 / 0.7) < 1e-5
    # the eight draws of a PAIR of 4-element groups come from ONE hash (four 32-bit words by a mixing chain, 16 bits per draw): every position keeps at the
    # nominal rate, and positions -- of one group and of neighbouring groups -- are pairwise independent
    big = torch.ones(8, 128, 128, 16, device=DEV)
    for p_drop in (0.05, 0.3, 0.5):
        zb = torch.empty_like(big)
        lib.bn_act_fwd(big, one, zero, zb, 1.0, (lib.DROP_RNG_ELEM, p_drop, 99, None, off))
        k = (zb != 0).float().view(-1, 2, 4)                 # [pair of groups][group][position]
        n = k.shape[0] * 2
        tol = 4.0 * (p_drop * (1 - p_drop) / n) ** 0.5 + 2.0 ** -15
        rate = k.mean(dim=(0, 1))
        assert bool(((rate - (1 - p_drop)).abs() < tol).all()), (p_drop, rate.tolist())
        flat = k.view(-1, 8)
        cov = (flat.t() @ flat) / flat.shape[0] - flat.mean(0)[:, None] * flat.mean(0)[None, :]
        cov.fill_diagonal_(0.0)
        assert float(cov.abs().max()) < 5.0 * p_drop * (1 - p_drop) / flat.shape[0] ** 0.5, (p_drop, float(cov.abs().max()))


@pytest.mark.parametrize("dtype", DTYPES)
def test_maxpool_and_upsample(dtype):
    lib = L()
    N, C, H, W = 2, 16, 12, 20
    x = q(rnd(N, C, H, W, seed=1), dtype)
    x[:, :, ::4, ::4] = 0.0
    x[:, :, 1::4, 1::4] = 0.0                         # ties inside windows (dropout zeros)
    x = x.requires_grad_(True)
    y = F.max_pool2d(x, 2)
    dy = q(rnd(N, C, H // 2, W // 2, seed=2), dtype)
    y.backward(dy)
    xd = nhwc(x.detach(), dtype)
    yd = torch.empty(N, H // 2, W // 2, C, dtype=dtype, device=DEV)
    lib.maxpool2_fwd(xd, yd)
    close(yd.permute(0, 3, 1, 2), y.detach(), dtype, "maxpool fwd")
    dxd = torch.empty_like(xd)
    lib.maxpool2_bwd(xd, nhwc(dy, dtype), dxd)
    close(dxd.permute(0, 3, 1, 2), x.grad, dtype, "maxpool bwd (first max wins)")
    for (h, w) in [(6, 10), (1, 1), (4, 4), (17, 3)]:
        u = q(rnd(N, C, h, w, seed=3), dtype).requires_grad_(True)
        up = F.interpolate(u, scale_factor=2, mode="bilinear", align_corners=True)
        g = q(rnd(N, C, 2 * h, 2 * w, seed=4), dtype)
        up.backward(g)
        ud = nhwc(u.detach(), dtype)
        upd = torch.empty(N, 2 * h, 2 * w, C, dtype=dtype, device=DEV)
        lib.upsample2x_fwd(ud, upd)
        close(upd.permute(0, 3, 1, 2), up.detach(), dtype, f"upsample fwd {h}x{w}")
        dud = torch.empty_like(ud)
        lib.upsample2x_bwd(nhwc(g, dtype), dud)
        close(dud.permute(0, 3, 1, 2), u.grad, dtype, f"upsample bwd {h}x{w}")


def test_ce_and_dice_against_golden(golden):
    from oracle import losses_ref
    lib = L()
    g = golden("g3_losses.npz")
    for C in (2, 3):
        logits = torch.tensor(g[f"logits{C}"])
        lab = torch.from_numpy(g[f"labels{C}"])
        ld = logits.permute(0, 2, 3, 1).contiguous().to(DEV)
        acc = torch.zeros(2 * lib.CE_SLOTS, dtype=torch.float64, device=DEV)
        lib.ce_fwd(ld, lab.to(DEV), C, acc)
        loss = torch.empty(1, device=DEV)
        lib.ce_finalize(acc, loss)
        assert abs(loss.item() - float(g[f"ce{C}"])) < 1e-5
        dl = torch.empty_like(ld)
        lib.ce_bwd(ld, lab.to(DEV), C, acc, None, dl)
        assert torch.allclose(dl.permute(0, 3, 1, 2).cpu(), torch.tensor(g[f"ce_grad{C}"]), atol=1e-8, rtol=1e-4)
        dense = torch.from_numpy(g[f"dense{C}"])[:, 0]
        counts = torch.zeros(3 * (C - 1), dtype=torch.int64, device=DEV)
        lib.dice_counts(ld, dense.to(DEV), counts)
        pred = logits.argmax(1).numpy()
        cnt = counts.cpu().numpy().reshape(C - 1, 3)
        for i in range(1, C):
            P = (pred == 1) if i == 1 else (pred >= 1)
            G = (dense.numpy() == 1) if i == 1 else (dense.numpy() >= 1)
            assert tuple(cnt[i - 1]) == (int((P & G).sum()), int(P.sum()), int(G.sum()))     # bit-exact integers
            d_dev = 2.0 * cnt[i - 1, 0] / (cnt[i - 1, 1] + cnt[i - 1, 2])
            assert d_dev == 2.0 * (P & G).sum() / (P.sum() + G.sum())
    # all-ignored -> NaN like torch
    acc = torch.zeros(2 * lib.CE_SLOTS, dtype=torch.float64, device=DEV)
    lib.ce_fwd(torch.zeros(1, 8, 8, 2, device=DEV), torch.full((1, 8, 8), 2, dtype=torch.uint8, device=DEV), 2, acc)
    loss = torch.empty(1, device=DEV)
    lib.ce_finalize(acc, loss)
    assert torch.isnan(loss).item()


def test_pdice_and_dice_losses_against_reference_golden(golden):
    """fedicra_amd.losses.{pDLoss, DiceLoss} vs values / gradients produced by the reference's own classes
    (incl. pDLoss's [B,B,H,W] mask broadcast)."""
    from fedicra_amd.losses import CrossEntropyLoss, DiceLoss, pDLoss
    g = golden("g3_losses.npz")
    for C in (2, 3):
        logits = torch.tensor(g[f"logits{C}"])
        lab = torch.from_numpy(g[f"labels{C}"]).to(DEV)
        probs = torch.softmax(logits, 1).to(DEV).requires_grad_(True)
        pd = pDLoss(C, ignore_index=C)(probs, lab.unsqueeze(1))
        pd.backward()
        assert abs(pd.item() - float(g[f"pdice{C}"])) < 2e-6, (pd.item(), float(g[f"pdice{C}"]))
        assert torch.allclose(probs.grad.cpu(), torch.tensor(g[f"pdice_grad{C}"]), rtol=1e-4, atol=1e-9)
        probs2 = torch.softmax(logits, 1).to(DEV).requires_grad_(True)
        dl = DiceLoss(C)(probs2, torch.from_numpy(g[f"dense{C}"]).to(DEV))
        dl.backward()
        assert abs(dl.item() - float(g[f"dice{C}"])) < 2e-6
        assert torch.allclose(probs2.grad.cpu(), torch.tensor(g[f"dice_grad{C}"]), rtol=1e-4, atol=1e-9)
        lg = logits.clone().to(DEV).requires_grad_(True)
        ce = CrossEntropyLoss(ignore_index=C)(lg, lab)
        ce.backward()
        assert abs(ce.item() - float(g[f"ce{C}"])) < 1e-5
        assert torch.allclose(lg.grad.cpu(), torch.tensor(g[f"ce_grad{C}"]), rtol=1e-4, atol=1e-8)
    # the single-site trainer's validation loss 0.5*(CE + pDLoss) (Unet_pCE.py:171-177) composes on the device
    w = [0.3, 0.7]
    probs = torch.softmax(torch.tensor(g["logits2"]), 1)
    ref_w = None
    from oracle.losses_ref import _one_hot
    t = torch.from_numpy(g["labels2"]).unsqueeze(1)
    mask = (t != 2).float()
    oh = _one_hot(t, 2)
    ref_w = sum(w[i] * (1 - (2 * (probs[:, i] * oh[:, i] * mask).sum() + 1e-5) /
                        ((probs[:, i] ** 2 * mask).sum() + (oh[:, i] ** 2 * mask).sum() + 1e-5)) for i in range(2)) / 2
    got = pDLoss(2, 2)(probs.to(DEV), t.to(DEV), weight=w)
    assert abs(got.item() - ref_w.item()) < 2e-6


def test_adamw_matches_torch():
    lib = L()
    n = 10_007
    p0, gs = rnd(n, seed=1), [rnd(n, seed=10 + i, scale=0.1) for i in range(4)]
    ref = p0.clone().requires_grad_(True)
    opt = torch.optim.AdamW([ref], lr=0.01, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2)
    p = p0.clone().to(DEV)
    m, v = torch.zeros(n, device=DEV), torch.zeros(n, device=DEV)
    step = torch.zeros(1, dtype=torch.int32, device=DEV)
    it = torch.zeros(1, dtype=torch.int32, device=DEV)
    lr = torch.tensor([0.01], dtype=torch.float64, device=DEV)
    hyper = torch.empty(4, device=DEV)
    shadow = torch.empty(n, dtype=torch.bfloat16, device=DEV)
    cur = 0.01
    for i, g in enumerate(gs):
        ref.grad = g.clone()
        for grp in opt.param_groups:
            grp["lr"] = cur
        opt.step()
        lib.adamw_hyper(step, hyper, lr, 0.9, 0.999, 1e-2)
        lib.adamw_step(p, g.to(DEV), m, v, hyper, 0.9, 0.999, 1e-8, shadow)
        lib.lr_poly_advance(it, lr, 0.01, 30000.0)
        cur = 0.01 * (1.0 - (i + 1) / 30000.0) ** 0.9
        assert abs(lr.item() - cur) < 1e-15
    assert int(step) == 4 and int(it) == 4
    assert torch.allclose(p.cpu(), ref.detach(), atol=2e-7, rtol=1e-6)
    assert torch.equal(shadow.cpu(), p.cpu().to(torch.bfloat16))


def test_scale_ala_pcs_helpers():
    lib = L()
    n = 5000
    w, gr, lo, gl = rnd(n, seed=1).abs(), rnd(n, seed=2), rnd(n, seed=3), rnd(n, seed=4)
    wd, td = w.clone().to(DEV), torch.empty(n, device=DEV)
    lib.ala_update(wd, td, gr.to(DEV), lo.to(DEV), gl.to(DEV), 1.0)
    wr = torch.clamp(w - 1.0 * (gr * (lo - gl)), 0, 1)
    assert torch.allclose(wd.cpu(), wr, atol=1e-7) and torch.allclose(td.cpu(), gl + (lo - gl) * wr, atol=1e-6)
    out = torch.empty(n, device=DEV)
    lib.scale(gr.to(DEV), out, 21.0 / 113.0)
    assert torch.allclose(out.cpu(), gr * (21.0 / 113.0), atol=1e-7)
    for dtype in DTYPES:
        N, C, H, W = 3, 256, 4, 5
        x = q(rnd(N, C, H, W, seed=5), dtype)
        xd = nhwc(x, dtype)
        avg, mx = torch.empty(N, C, device=DEV), torch.empty(N, C, device=DEV)
        am = torch.empty(N, C, dtype=torch.int32, device=DEV)
        lib.global_avgmax(xd, avg, mx, am)
        assert torch.allclose(avg.cpu(), x.mean((2, 3)), atol=1e-5) and torch.equal(mx.cpu(), x.amax((2, 3)))
        assert torch.equal(am.cpu().long(), x.reshape(N, C, -1).argmax(2))
        h = torch.sigmoid(rnd(N, C, seed=6))
        y = torch.empty_like(xd)
        lib.channel_gate_fwd(xd, h.to(DEV), y)
        close(y.permute(0, 3, 1, 2), x * h[:, :, None, None] + x, dtype, "gate fwd")


def test_layout_and_cast():
    lib = L()
    x = rnd(2, 3, 5, 7, seed=1)
    for dtype in DTYPES:
        d = torch.empty(2, 5, 7, 3, dtype=dtype, device=DEV)
        lib.nchw_to_nhwc(x.to(DEV), d)
        assert torch.equal(d.cpu().float(), x.permute(0, 2, 3, 1).to(dtype).float())
        back = torch.empty(2, 3, 5, 7, device=DEV)
        lib.nhwc_to_nchw(d, back)
        assert torch.equal(back.cpu(), x.to(dtype).float())


def test_hd95_surface_distances_match_scipy_restatement():
    """hd95 (a17): device border extraction + exhaustive nearest-border search against the scipy restatement of medpy's
    algorithm (oracle.losses_ref.hd95_percase) -- blobs, nested classes, a mask touching the image edge, speckle and the
    empty cases."""
    from fedicra_amd.flower_common import hd95_table
    from oracle.losses_ref import hd95_percase
    rng = np.random.default_rng(5)
    H, W, C = 48, 64, 3
    yy, xx = np.mgrid[0:H, 0:W]

    def disks(cx, cy, r1, r2):
        d = np.hypot(xx - cx, yy - cy)
        return (d < r1).astype(np.uint8) + (d < r2).astype(np.uint8)       # 0 / 1 ring / 2 core

    preds = [disks(30, 20, 14, 6), disks(5, 5, 12, 4), (rng.random((H, W)) < 0.3).astype(np.uint8), np.zeros((H, W), np.uint8),
             disks(40, 30, 9, 0)]
    gts = [disks(33, 22, 12, 7), disks(8, 4, 10, 5), disks(30, 20, 10, 3), disks(30, 20, 10, 3), np.zeros((H, W), np.uint8)]
    logits = torch.zeros(len(preds), C, H, W)
    for i, p in enumerate(preds):
        logits[i] = torch.nn.functional.one_hot(torch.from_numpy(p.astype(np.int64)), C).permute(2, 0, 1).float() * 4.0
    got = hd95_table(logits.to(DEV), torch.from_numpy(np.stack(gts)).to(DEV), C)
    for i, (p, g) in enumerate(zip(preds, gts)):
        for k in range(1, C):
            P = (p == 1) if k == 1 else (p >= 1)
            G = (g == 1) if k == 1 else (g >= 1)
            ref = hd95_percase(P, G)
            if np.isnan(ref):
                assert np.isnan(got[i, k - 1]), (i, k, got[i, k - 1])
            else:
                assert got[i, k - 1] == ref, (i, k, got[i, k - 1], ref)       # exact: integer d^2, one fp64 sqrt
