"""BASELINE.json configs[4] -- "8 non-IID multi-site clients, 3D U-Net + per-client adapter heads, fp16": the model
(fedicra_amd.networks.unet_3D.unet_3D_lc), its parity against the CPU restatement (oracle.unet3d_ref.RefUNet3DLC) in fp32
and in the fp16 compute mode, a three-step fp16 training loop under the GradScaler, and the odd-volume behaviour of the
reference's skip padding (networks/utils.py:271-274)."""
import argparse

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _pair(K=4, cid=2, seed=2040, ncls=2, in_ch=1):
    from fedicra_amd.networks.unet_3D import unet_3D_lc
    from oracle.unet3d_ref import RefUNet3DLC
    from oracle.unet_ref import seeded_state
    m = unet_3D_lc(n_classes=ncls, in_channels=in_ch, client_num=K, client_id=cid)
    r = RefUNet3DLC(n_classes=ncls, in_channels=in_ch, client_num=K, client_id=cid)
    assert list(m.state_dict().keys()) == list(r.state_dict().keys())
    for mod in (m, r):
        extra = {f"pcs_list.{i}.{k}": v for i, p in enumerate(mod.pcs_list) for k, v in p.state_dict().items()}
        seeded_state(mod, seed, extra=extra)
    return m.cuda(), r


def _data(seed=7, B=2, S=32, ncls=2):
    rng = np.random.default_rng(seed)
    x = torch.from_numpy(rng.random((B, 1, S, S, S), dtype=np.float32))
    zz, yy, xx = np.mgrid[0:S, 0:S, 0:S]
    lab = ((zz - S / 2) ** 2 + (yy - S / 2) ** 2 + (xx - S / 2) ** 2 < (S / 4) ** 2).astype(np.int64)
    lab = np.broadcast_to(lab, (B, S, S, S)).copy()
    lab[rng.random(lab.shape) < 0.8] = ncls                   # scribble-like: 80 % unlabeled (ignore index)
    return x, torch.from_numpy(lab)


@pytest.mark.parametrize("dtype", ["fp32", "fp16"])
def test_unet3d_lc_forward_and_backward_against_the_cpu_restatement(dtype):
    """Logits, site heat-map (own and another client's embedding, incl. the `emb_idx = 0 means own` rule), auxiliary head
    and every parameter gradient of pCE(logits) + 0.4 * pCE(aux) - MSE(heat-maps), at 32^3 (eval mode: InstanceNorm has no
    running state, dropout is the only train / eval difference).  fp32: the parity mode; fp16: configs[4]'s compute mode
    (16-bit storage and MFMA operands, fp32 accumulation / statistics / loss)."""
    m, r = _pair()
    m.set_compute_dtype(dtype).eval()
    r.eval()
    x, lab = _data()
    o_own, o_oth, o_zero = r(x), r(x, 3), r(x, 0)
    assert torch.equal(o_zero[6][-1], o_own[6][-1]) and not torch.equal(o_oth[6][-1], o_own[6][-1])
    up = F.interpolate(lab[:, None].float(), size=o_own[7].shape[2:], mode="nearest")[:, 0].long()
    loss_r = F.cross_entropy(o_own[0], lab, ignore_index=2) + 0.4 * F.cross_entropy(o_own[7], up, ignore_index=2) \
        - F.mse_loss(o_own[6][-1], o_oth[6][-1].detach())
    loss_r.backward()
    g_own = m(x.to(DEV))
    with torch.no_grad():
        g_oth, g_zero = m(x.to(DEV), 3), m(x.to(DEV), 0)
    assert len(g_own) == 8 and len(g_own[1]) == 5 and g_own[6][-1].shape == (2, 256, 1, 1, 1)
    tol = {"fp32": 2e-4, "fp16": 3e-2}[dtype]
    for i in (0, 7):
        err = (g_own[i].detach().float().cpu() - o_own[i].detach()).abs().max().item()
        assert err < tol * max(1.0, o_own[i].abs().max().item()), (i, err)
    for a, b in ((g_own[6][-1], o_own[6][-1]), (g_oth[6][-1], o_oth[6][-1])):
        assert (a.detach().float().cpu() - b.detach()).abs().max().item() < (1e-5 if dtype == "fp32" else 5e-3)
    assert torch.equal(g_zero[6][-1], g_own[6][-1].detach())
    upd = up.to(DEV)
    loss = F.cross_entropy(g_own[0], lab.to(DEV), ignore_index=2) + 0.4 * F.cross_entropy(g_own[7], upd, ignore_index=2) \
        - F.mse_loss(g_own[6][-1].float(), g_oth[6][-1].float().detach())
    assert abs(loss.item() - loss_r.item()) < (2e-5 if dtype == "fp32" else 5e-3)
    loss.backward()
    ref_g = {n: p.grad for n, p in r.named_parameters()}
    worst = 0.0
    for n, p in m.named_parameters():
        if n.endswith(".0.bias"):
            continue                                  # bias in front of an InstanceNorm: true gradient 0
        a, b = p.grad.float().cpu(), ref_g[n]
        rel = (a - b).abs().max().item() / max(b.abs().max().item(), 1e-8)
        worst = max(worst, rel)
        # ReLU'(v) jumps at 0: an InstanceNorm output at round-off distance from 0 (some always are among 10^5..10^6) flips
        # a whole gradient path between two correct fp32 implementations -- the bound is that of the 2D full-size tests
        if dtype == "fp32":
            assert rel < 3e-2, (n, rel)
        else:
            # 16-bit storage through five levels down to 2^3 voxels per instance (InstanceNorm over 8 values): element-wise
            # bounds are meaningless there; direction and size of every gradient tensor must still agree
            cos = float(torch.nn.functional.cosine_similarity(a.flatten(), b.flatten(), dim=0))
            ratio = float(a.norm() / b.norm().clamp_min(1e-12))
            assert cos > 0.97 and 0.9 < ratio < 1.1, (n, cos, ratio)
    print(f"unet_3D_lc {dtype}: worst relative gradient error {worst:.2e}")


def test_unet3d_lc_fp16_training_loop_with_gradscaler():
    """configs[4] end to end on one client: three fp16 iterations of forward, pCE, backward, GradScaler (scale, unscale,
    inf check), fused AdamW over the flat state of the 3D model (FlatStoreMixin: the same aggregation / ALA plumbing as the
    2D models), poly LR -- the loss falls, nothing overflows at the default 2^16 scale, and only parameters that received a
    gradient moved (the frozen, unregistered channel selection did not)."""
    from fedicra_amd import ops
    from fedicra_amd.amp import GradScaler
    from fedicra_amd.optim import FusedAdamW
    m, _ = _pair(seed=2041)
    m.set_compute_dtype("fp16").train()
    x, lab = _data(seed=9)
    x, lab = x.to(DEV), lab.to(DEV)
    assert m.flat_params.numel() > 5_000_000 and m.flat_state.numel() == m.flat_params.numel()      # no float buffers
    pcs_before = [q.detach().clone() for q in m.pcs_list[0].parameters()]
    first = m.flat_params.clone()
    opt = FusedAdamW(m, lr=0.01, base_lr=0.01, max_iterations=100)
    sc = GradScaler()
    ops.manual_seed(5)
    losses = []
    for it in range(3):
        ops.begin_iteration(x.device)
        opt.zero_grad()
        out = m(x)
        loss = F.cross_entropy(out[0], lab, ignore_index=2)
        sc.scale(loss).backward()
        sc.step(opt)
        sc.update()
        opt.advance_lr()
        losses.append(float(loss.detach()))
    assert np.isfinite(losses).all() and losses[-1] < losses[0], losses
    assert sc.get_scale() == 65536.0 and int(opt.steps[0].item()) == 3
    assert not torch.equal(first, m.flat_params)
    assert all(torch.equal(a, b) for a, b in zip(pcs_before, m.pcs_list[0].parameters()))
    names = {n for n, p in m.named_parameters() if p.grad is not None}
    assert "final.weight" in names and "conv1.conv1.0.weight" in names and not any(n.startswith("dsn_head") for n in names)
    assert [k for k in m.LOCAL_KEYS if any(k in n for n in names)] == list(m.LOCAL_KEYS)


def test_odd_volumes_fail_like_the_references_padding_rule():
    """networks/utils.py:271-274 pads the skip tensor by 2 * [offset // 2, offset // 2, 0]; for the only non-zero offset a
    floor-pooled encoder produces (-1) that crops W twice and the concatenation raises -- in the restated reference and,
    with the same error class, here."""
    from fedicra_amd.networks.unet_3D import unet_3D
    from oracle.unet3d_ref import RefUNet3D
    x = torch.rand(1, 1, 40, 40, 40)                            # 40 -> 20 -> 10 -> 5 -> 2: the 5^3 skip meets a 4^3 up-sampling
    with pytest.raises(RuntimeError, match="Sizes of tensors must match"):
        RefUNet3D(n_classes=2, in_channels=1).eval()(x)
    with pytest.raises(RuntimeError, match="Sizes of tensors must match"):
        unet_3D(n_classes=2, in_channels=1).cuda().eval()(x.to(DEV))
    assert unet_3D(n_classes=2, in_channels=1).cuda().eval()(torch.rand(1, 1, 16, 32, 48).to(DEV)).shape == (1, 2, 16, 32, 48)
