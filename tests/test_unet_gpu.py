"""GPU parity of the HIP-backed U-Net family and training loop against (a) the golden vectors the
reference itself produced (tests/golden) and (b) the CPU oracle run side by side on the same
seeded inputs.  Tolerances: fp32 mode -- logits / Dice inputs within 1e-4 (north_star); bf16 mode --
reported, loose."""
import argparse

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _mask_provider():
    """Dropout keep-masks drawn exactly like nn.Dropout on the CPU (same generator stream as the oracle)."""
    return lambda shape, p: torch.empty(shape).bernoulli_(1 - p)


def _mk(cls, *a, dtype="fp32", seed=2022, lc=False):
    from fedicra_amd.networks.unet import set_compute_dtype
    from oracle.unet_ref import seeded_state
    m = cls(*a)
    extra = None
    if lc:
        extra = {f"encoder.pcs_list.{i}.{k}": v for i, p in enumerate(m.encoder.pcs_list)
                 for k, v in p.state_dict().items()}
    seeded_state(m, seed, extra=extra)
    m = m.cuda()
    set_compute_dtype(m, dtype)
    return m


def test_unet_eval_logits_match_reference_golden(golden):
    from fedicra_amd.networks.unet import UNet
    from helpers import assert_ck
    g = golden("g2_unet_fwd.npz")
    m = _mk(UNet, 1, 2).eval()
    x = torch.from_numpy(g["x"]).unsqueeze(1).to(DEV)
    with torch.no_grad():
        o = m(x)
    assert len(o) == 6 and len(o[1]) == 5
    err = (o[0].cpu() - torch.from_numpy(g["eval_logits"])).abs().max().item()
    assert err < 1e-4, f"eval logits max err {err:.3e}"
    assert o[0].shape == (4, 2, 64, 64) and o[1][4].shape == (4, 256, 4, 4)
    for i, f in enumerate(o[1]):
        assert_ck(f.float().cpu(), g[f"eval_feat{i}_ck"], rtol=2e-5, atol=1e-5, what=f"feat{i}")
    for i in range(2, 6):
        assert_ck(o[i].float().cpu(), g[f"eval_de{i-1}_ck"], rtol=2e-5, atol=1e-5, what=f"de{i-1}")
    m3 = _mk(UNet, 3, 3, seed=2023).eval()
    with torch.no_grad():
        l3 = m3(torch.from_numpy(g["x3"]).to(DEV))[0]
    assert (l3.cpu() - torch.from_numpy(g["eval_logits3"])).abs().max().item() < 1e-4


def test_unet_train_forward_with_reference_masks(golden):
    from fedicra_amd import ops
    from fedicra_amd.networks.unet import UNet
    from helpers import assert_ck
    g = golden("g2_unet_fwd.npz")
    m = _mk(UNet, 1, 2).train()
    x = torch.from_numpy(g["x"]).unsqueeze(1).to(DEV)
    ops.set_dropout_mask_provider(_mask_provider())
    try:
        torch.manual_seed(7)
        o = m(x)
    finally:
        ops.set_dropout_mask_provider(None)
    err = (o[0].detach().cpu() - torch.from_numpy(g["train_logits_seed7"])).abs().max().item()
    assert err < 1e-4, f"train-mode logits max err {err:.3e}"
    for k, v in m.state_dict().items():
        if "running" in k or "num_batches" in k:
            assert_ck(v.double().cpu(), g["after_train_fwd/" + k], rtol=1e-5, atol=1e-6, what=k)


def test_unet_lc_forward_and_quirks(golden):
    from fedicra_amd.networks.unet import UNet_LC, UNet_LC_MultiHead
    from helpers import assert_ck
    g = golden("g2_unet_lc_fwd.npz")
    m = _mk(UNet_LC, 1, 2, 1, 8, 8, 3, lc=True).eval()
    assert not any("pcs" in k for k in m.state_dict()) and len(m.state_dict()) == 144
    x = torch.from_numpy(g["x"]).unsqueeze(1).to(DEV)
    with torch.no_grad():
        for e in (None, 0, 5):
            o = m(x, e)
            assert len(o) == 8
            assert (o[0].cpu() - torch.from_numpy(g[f"eval_logits_e{e}"])).abs().max().item() < 1e-4
            assert (o[6][-1].cpu() - torch.from_numpy(g[f"eval_hmap_e{e}"])).abs().max().item() < 1e-5
            assert (o[7].cpu() - torch.from_numpy(g[f"eval_aux_e{e}"])).abs().max().item() < 1e-4
            assert o[6][:4] == [None] * 4 and o[6][-1].shape == (4, 256, 1, 1)
    mh = _mk(UNet_LC_MultiHead, 1, 2, 1, 8, 8, 2, seed=2024, lc=True).eval()
    with torch.no_grad():
        o = mh(x)
    assert len(o) == 10 and len(mh.state_dict()) == 160
    assert (o[0].cpu() - torch.from_numpy(g["mh_eval_logits"])).abs().max().item() < 1e-4
    for i in (7, 8, 9):
        assert_ck(o[i].float().cpu(), g[f"mh_eval_aux{i-6}_ck"], rtol=2e-5, atol=1e-5, what=f"aux{i-6}")


@pytest.mark.parametrize("dtype,bound", [("bf16", 0.15), ("fp16", 0.02)])
def test_unet_bf16_mode_close_to_fp32(golden, dtype, bound):
    """16-bit storage modes against the reference's fp32 logits: bf16 (8 significant bits) and fp16 (11)."""
    from fedicra_amd.networks.unet import UNet
    g = golden("g2_unet_fwd.npz")
    m = _mk(UNet, 1, 2, dtype=dtype).eval()
    x = torch.from_numpy(g["x"]).unsqueeze(1).to(DEV)
    with torch.no_grad():
        lg = m(x)[0]
    ref = torch.from_numpy(g["eval_logits"])
    assert lg.dtype == torch.float32
    err = (lg.cpu() - ref).abs().max().item()
    agree = (lg.cpu().argmax(1) == ref.argmax(1)).float().mean().item()
    print(f"{dtype} eval logits: max err {err:.3e}, argmax agreement {agree:.5f}")
    assert err < bound and agree > 0.99


def _args(**kw):
    a = argparse.Namespace(strategy="FedAvg", amp=0, model="unet", cid=0, min_num_clients=1, num_classes=2,
                           img_class="faz", base_lr=0.01, max_iterations=30000, iters=5, rep_iters=3, alpha=0.5,
                           snapshot_path=None, use_graph=False)
    a.__dict__.update(kw)
    return a


def test_backward_matches_oracle_grads():
    """One fwd+bwd, dropout masks pinned: every parameter gradient vs the CPU oracle."""
    from fedicra_amd import ops
    from fedicra_amd.networks.unet import UNet
    from oracle.losses_ref import pce_loss
    from oracle.unet_ref import RefUNet, seeded_state
    from helpers import loader
    b = loader(1, 4, 64, cid=0)[0]
    ref = RefUNet(1, 2)
    seeded_state(ref, 2022)
    ref.train()
    torch.manual_seed(3)
    pce_loss(ref(b["image"].unsqueeze(1))[0], b["label"], 2).backward()
    m = _mk(UNet, 1, 2).train()
    ops.set_dropout_mask_provider(_mask_provider())
    try:
        torch.manual_seed(3)
        out = m(b["image"].unsqueeze(1).to(DEV))
        loss = ops.ce_loss(out[0].permute(0, 2, 3, 1), b["label"].to(DEV), 2)
        loss.backward()
    finally:
        ops.set_dropout_mask_provider(None)
    worst = 0.0
    for (n, p), (_, q) in zip(m.named_parameters(), ref.named_parameters()):
        assert p.grad is not None, n
        gq = q.grad
        scale = max(gq.abs().max().item(), 1e-6)
        if n.endswith("conv_conv.0.bias") or n.endswith("conv_conv.4.bias"):
            continue        # conv bias before BN: true gradient is 0, both sides hold only round-off
        e = (p.grad.cpu() - gq).abs().max().item() / scale
        worst = max(worst, e)
        assert e < 2e-3, f"{n}: rel grad err {e:.3e}"
    print(f"worst relative parameter-gradient error {worst:.3e}")


def test_pcs_module_forward_backward_vs_cpu():
    from fedicra_amd.networks.unet import PersonalizedChannelSelection
    from oracle.unet_ref import RefPCS
    torch.manual_seed(0)
    ref = RefPCS(256, 3)
    pcs = PersonalizedChannelSelection(256, 3)
    pcs.load_state_dict(ref.state_dict())
    pcs = pcs.cuda()
    x = torch.randn(4, 256, 4, 4)
    emb = torch.zeros(4, 3)
    emb[:, 1] = 1
    xr = x.clone().requires_grad_(True)
    yr, hr = ref(xr, emb)
    gy, gh = torch.randn_like(yr), torch.randn_like(hr)
    (yr * gy).sum().add((hr * gh).sum()).backward()
    xd = x.clone().to(DEV).requires_grad_(True)
    yd, hd = pcs(xd, emb.to(DEV))
    ((yd * gy.to(DEV)).sum() + (hd * gh.to(DEV)).sum()).backward()
    assert (yd.detach().cpu() - yr.detach()).abs().max().item() < 1e-5
    assert (hd.detach().cpu() - hr.detach()).abs().max().item() < 1e-6
    e = (xd.grad.cpu() - xr.grad).abs().max().item() / xr.grad.abs().max().item()
    assert e < 1e-5, f"PCS dx rel err {e:.3e}"


@pytest.mark.parametrize("phase", ["head", "body"])
def test_lc_backward_matches_oracle_grads(phase):
    """FedICRA loss (CE + alpha * LC, K=3) on UNet_LC, masks pinned: every parameter gradient vs the oracle."""
    from fedicra_amd import ops
    from fedicra_amd.networks.unet import UNet_LC
    from oracle.losses_ref import pce_loss
    from oracle.unet_ref import RefUNetLC, pcs_named_tensors, seeded_state
    from helpers import loader
    K, cid = 3, 1
    b = loader(1, 4, 64, cid=cid)[0]
    head = ("decoder.out_conv.weight", "decoder.out_conv.bias")

    def fed_loss(model, x, y, dev_ce):
        out = model(x)
        ce = dev_ce(out[0], y)
        acc = 0
        for other in range(K):
            if other == cid:
                continue
            with torch.no_grad():
                ho = model(x, other)[6][-1]
            acc = acc + torch.nn.functional.mse_loss(out[6][-1], ho.detach())
        return ce + 1.0 * (-acc / (K - 1))

    def mkref():
        r = RefUNetLC(1, 2, 1, K, K, cid)
        seeded_state(r, 2022, extra=pcs_named_tensors(r))
        return r

    from helpers import clean_seed
    seed = clean_seed(mkref, b["image"].unsqueeze(1))
    ref = mkref()
    ref.train()
    for n, p in ref.named_parameters():
        p.requires_grad = (n in head) == (phase == "head")
    torch.manual_seed(seed)
    lr_ = fed_loss(ref, b["image"].unsqueeze(1), b["label"], lambda lg, y: pce_loss(lg, y, 2))
    lr_.backward()
    m = _mk(UNet_LC, 1, 2, 1, K, K, cid, lc=True).train()
    for n, p in m.named_parameters():
        p.requires_grad = (n in head) == (phase == "head")
    ops.set_dropout_mask_provider(_mask_provider())
    try:
        torch.manual_seed(seed)
        ld = fed_loss(m, b["image"].unsqueeze(1).to(DEV), b["label"].to(DEV),
                      lambda lg, y: ops.ce_loss(lg.permute(0, 2, 3, 1), y, 2))
        ld.backward()
    finally:
        ops.set_dropout_mask_provider(None)
    assert abs(ld.item() - lr_.item()) < 1e-5
    worst, worst_n, errs = 0.0, "", []
    for (n, p), (_, q) in zip(m.named_parameters(), ref.named_parameters()):
        assert (p.grad is None) == (q.grad is None), n
        if q.grad is None:
            continue
        if n.endswith("conv_conv.0.bias") or n.endswith("conv_conv.4.bias") or n.endswith("dsn_head.0.bias"):
            continue
        scale = max(q.grad.abs().max().item(), 1e-6)
        e = (p.grad.cpu() - q.grad).abs().max().item() / scale
        errs.append(e)
        if n.endswith("weight") and ("conv_conv.0" in n or "conv_conv.4" in n or "out_conv" in n or "conv1x1" in n):
            print(f"   {n:55s} rel err {e:.2e}")
        if e > worst:
            worst, worst_n = e, n
    errs = np.array(errs)
    print(f"[{phase}] parameter-gradient rel err: median {np.median(errs):.3e}, worst {worst:.3e} at {worst_n}")
    # clean_seed() excluded draws with a pre-activation at round-off distance from 0 (LeakyReLU' jumps there)
    assert np.median(errs) < 2e-5 and worst < 1e-3, (worst, worst_n, seed)


@pytest.mark.parametrize("use_graph", [False, True])
def test_local_train_matches_reference_trajectory(golden, use_graph):
    """MyClient._train, 5 iterations + a second round of 2, vs the reference's own _train (golden g4)."""
    from fedicra_amd import ops
    from fedicra_amd.flower_common import MyModel
    from fedicra_amd.flower_pCE_2D import MyClient
    from fedicra_amd.networks.unet import UNet
    from helpers import assert_ck, loader
    g = golden("g4_train_unet.npz")
    args = _args(use_graph=use_graph)
    batches = loader(3, 4, 64, cid=0)
    net = _mk(UNet, 1, 2)
    client = MyClient(args, MyModel(args, net, batches, batches), batches, batches)
    if use_graph:
        # graph replay cannot call a host mask provider: use masks recorded per iteration instead
        pytest.skip("dropout masks come from a host callback in parity mode; graph path is covered by "
                    "test_graph_replay_equals_eager")
    ops.set_dropout_mask_provider(_mask_provider())
    try:
        torch.manual_seed(2022)
        last, met = client._train({"iter_global": 5, "iters": 5, "eval_iters": 10, "batch_size": 4, "stage": "fit"})
        errs = np.abs(np.array(client.last_losses) - g["losses_6dp"])
        print("loss trajectory", client.last_losses, "ref", g["losses_6dp"].tolist(), "err", errs.tolist())
        # AdamW's first steps are sign-like (g / (|g| + 1e-8)), which makes the REFERENCE trajectory itself
        # round-off chaotic: the oracle run with 1 CPU thread instead of 8 already moves the losses by
        # [6e-8, 6e-5, 4e-3, 7e-4, 1e-3] (DESIGN.md "parity bar").  Step 0 is exact-fp32 parity; later steps
        # are held to that measured sensitivity.
        assert errs[0] < 1e-5 and errs[1] < 5e-4 and errs[2:].max() < 5e-3, f"loss trajectory err {errs}"
        assert abs(client.current_lr - float(g["lr_after"])) < 1e-15
        sd = net.state_dict()
        ow = torch.from_numpy(g["out_conv_weight"])
        assert (sd["decoder.out_conv.weight"].cpu() - ow).abs().max().item() < 2e-2
        args.iters = 2
        last2, _ = client._train({"iter_global": 7, "iters": 2, "eval_iters": 10, "batch_size": 4, "stage": "fit"})
        errs2 = np.abs(np.array(client.last_losses) - g["losses_round2_6dp"])
        # 6-7 AdamW steps in, the round-off chaos above has grown to the size of the loss differences between
        # runs of the reference itself; round 2 checks the mechanics (fresh AdamW, carried lr / iteration
        # counter, BN counters below), not the digits.
        assert np.isfinite(client.last_losses).all() and errs2.max() < 0.1, f"round-2 loss err {errs2}"
    finally:
        ops.set_dropout_mask_provider(None)
    nbt = [v for k, v in sd.items() if k.endswith("num_batches_tracked")]
    assert all(int(v) == 7 for v in nbt)


def test_single_adamw_step_matches_oracle_elementwise():
    """One full iteration from identical state and masks: every parameter element must match the oracle
    except where AdamW's first step is a coin flip (|grad| at round-off level -> update = +-lr)."""
    from fedicra_amd import ops
    from fedicra_amd.flower_common import MyModel
    from fedicra_amd.flower_pCE_2D import MyClient
    from fedicra_amd.networks.unet import UNet
    from oracle import fed_ref
    from oracle.unet_ref import RefUNet, seeded_state
    from helpers import loader
    batches = loader(1, 4, 64, cid=0)

    def mkref():
        r = RefUNet(1, 2)
        seeded_state(r, 2022)
        return r

    from helpers import clean_seed
    seed = clean_seed(mkref, batches[0]["image"].unsqueeze(1))
    ref = mkref()
    st = fed_ref.TrainState(0.01)
    torch.manual_seed(seed)
    fed_ref.local_train(ref, st, batches, iters=1, num_classes=2, base_lr=0.01, max_iterations=30000)
    args = _args(iters=1)
    net = _mk(UNet, 1, 2)
    client = MyClient(args, MyModel(args, net, batches, batches), batches, batches)
    ops.set_dropout_mask_provider(_mask_provider())
    try:
        torch.manual_seed(seed)
        client._train({"iter_global": 1, "iters": 1, "eval_iters": 10, "batch_size": 4, "stage": "fit"})
    finally:
        ops.set_dropout_mask_provider(None)
    tot = bad = 0
    gglob = max(q.grad.abs().max().item() for q in ref.parameters())
    for (n, p), (_, q) in zip(net.named_parameters(), ref.named_parameters()):
        d = (p.detach().cpu() - q.detach()).abs()
        flip = d > 1e-4
        # a mismatch is only legitimate where the oracle's own gradient is round-off sized (e.g. every conv
        # bias in front of a BatchNorm: its true gradient is exactly 0)
        # ... or sits in AdamW's non-linear zone g/(|g|+1e-8) with |g| within ~300x of eps, where an absolute
        # gradient difference of 1e-9 already moves the update by more than 1e-4.
        legit = q.grad.abs() < max(1e-5 * gglob, 3e-6)
        off = flip & ~legit
        assert not off.any(), (f"{n}: {int(off.sum())} elements differ with a real gradient: "
                               f"ref grads {q.grad[off][:5].tolist()} diffs {d[off][:5].tolist()} gmax {gglob:.3e}")
        tot += d.numel()
        bad += int(flip.sum())
    print(f"single step: {bad}/{tot} elements took the other sign of a round-off gradient")
    for (k, v), (_, w) in zip(net.state_dict().items(), ref.state_dict().items()):
        if "running" in k:
            assert torch.allclose(v.cpu(), w, atol=1e-5, rtol=1e-5), k


def test_graph_replay_equals_eager():
    """The captured-hipGraph iteration must produce the same parameters as eager launches (dropout RNG is
    device-side and keyed by the device iteration counter, so both paths draw identical masks)."""
    from fedicra_amd import ops
    from fedicra_amd.flower_common import MyModel
    from fedicra_amd.flower_pCE_2D import MyClient
    from fedicra_amd.networks.unet import UNet
    from helpers import loader
    batches = loader(3, 4, 64, cid=0)
    finals, losses = [], []
    for use_graph in (False, True):
        args = _args(use_graph=use_graph, iters=6)
        ops.manual_seed(1)
        net = _mk(UNet, 1, 2)
        client = MyClient(args, MyModel(args, net, batches, batches), batches, batches)
        client._train({"iter_global": 6, "iters": 6, "eval_iters": 12, "batch_size": 4, "stage": "fit"})
        finals.append(net.flat_state.clone())
        losses.append(list(client.last_losses))
    print("eager", losses[0], "graph", losses[1])
    # identical masks and kernels; only the fp32 atomic accumulation order (wgrad, BN sums) may differ
    assert np.allclose(losses[0][:2], losses[1][:2], atol=1e-6)
    assert np.allclose(losses[0], losses[1], atol=5e-3)


def test_fedicra_local_train_matches_reference(golden):
    from fedicra_amd import ops
    from fedicra_amd.flower_common import MyModel
    from fedicra_amd.flower_pCE_2D import MyClient
    from fedicra_amd.networks.unet import UNet_LC
    from helpers import loader
    g = golden("g5_fedicra_train.npz")
    K, cid = 3, 1
    args = _args(strategy="FedICRA", model="unet_lc", cid=cid, min_num_clients=K, iters=5, rep_iters=2, alpha=1.0)
    batches = loader(2, 4, 64, cid=cid)
    net = _mk(UNet_LC, 1, 2, 1, K, K, cid, lc=True)
    client = MyClient(args, MyModel(args, net, batches, batches), batches, batches)
    ops.set_dropout_mask_provider(_mask_provider())
    try:
        torch.manual_seed(2022)
        last, met = client._train({"iter_global": 60, "iters": 5, "eval_iters": 10, "batch_size": 4, "stage": "fit"})
    finally:
        ops.set_dropout_mask_provider(None)
    errs = np.abs(np.array(client.last_losses) - g["losses_6dp"])
    print("fedicra losses", client.last_losses, "ref", g["losses_6dp"].tolist(), "err", errs.tolist())
    assert errs[:4].max() < 1e-5        # head phase (only out_conv trains) + first body forward: fp32 parity
    assert errs[4] < 5e-3               # after the first sign-like AdamW step on the body (see trajectory test)
    assert abs(met[f"client_{cid}_loss_lc"] - float(g["loss_lc_last"])) < 1e-3
    ow = torch.from_numpy(g["out_conv_weight"])
    assert (net.state_dict()["decoder.out_conv.weight"].cpu() - ow).abs().max().item() < 1e-5


def test_ala_set_weights_matches_reference(golden):
    from fedicra_amd.flower_common import MyModel
    from fedicra_amd.networks.unet import UNet_LC
    from oracle.unet_ref import RefUNetLC, seeded_state
    from oracle import fed_ref
    from helpers import assert_ck, loader
    g = golden("g7_ala.npz")
    K, cid = 3, 1
    args = _args(strategy="FedICRA", model="unet_lc", cid=cid, min_num_clients=K)
    batches = loader(3, 4, 64, cid=cid)
    net = _mk(UNet_LC, 1, 2, 1, K, K, cid, seed=100, lc=True).eval()     # eval mode: deterministic
    donor = RefUNetLC(1, 2, 1, K, K, cid)
    seeded_state(donor, 200)
    glob = fed_ref.get_weights(donor)
    model = MyModel(args, net, batches, batches)
    model.eval()
    model.set_weights(glob, {"iter_global": 50})
    assert model.start_phase is True
    assert_ck(net.state_dict()["decoder.out_conv.weight"].cpu(), g["eval/skip50_out_conv_ck"], rtol=1e-6, atol=1e-7)
    net2 = _mk(UNet_LC, 1, 2, 1, K, K, cid, seed=100, lc=True).eval()
    model = MyModel(args, net2, batches, batches)
    model.eval()
    model.set_weights(glob, {"iter_global": 60})
    assert model.start_phase is False
    assert len(model.ala_epoch_losses) == int(g["eval/first_epochs"])
    ow = torch.from_numpy(g["eval/first_out_conv_weight"])
    err = (net2.state_dict()["decoder.out_conv.weight"].cpu() - ow).abs().max().item()
    assert err < 1e-4, f"ALA-mixed out_conv err {err:.3e}"
    for k, v in net2.state_dict().items():
        assert_ck(v.double().cpu(), g["eval/first/" + k], rtol=2e-4, atol=2e-5, what=k)
    seeded_state(donor, 300)
    glob2 = fed_ref.get_weights(donor)
    model.set_weights(glob2, {"iter_global": 70})
    assert len(model.ala_epoch_losses) == 1
    for k, v in net2.state_dict().items():
        assert_ck(v.double().cpu(), g["eval/second/" + k], rtol=2e-4, atol=2e-5, what=k)
    model.set_weights(glob2, {"iter_global": 80})            # identical global again -> early-out, all global
    assert_ck(net2.state_dict()["decoder.out_conv.weight"].cpu(), g["eval/third_out_conv_ck"], rtol=1e-6, atol=1e-7)


def test_evaluate_dice_matches_oracle():
    from fedicra_amd.flower_common import evaluate
    from fedicra_amd.networks.unet import UNet
    from fedicra_amd.synth import phantom_batch
    from oracle.losses_ref import eval_case
    from oracle.unet_ref import RefUNet, seeded_state
    args = _args()
    imgs, _, dense = phantom_batch(6, 64, 1, 2, cid=2, dense=True)
    val = [{"image": torch.from_numpy(imgs[i:i + 1]), "label": torch.from_numpy(dense[i:i + 1])} for i in range(6)]
    net = _mk(UNet, 1, 2, seed=5)
    met = evaluate(args, net, val)
    ref = RefUNet(1, 2)
    seeded_state(ref, 5)
    ref.eval()
    tot = 0.0
    with torch.no_grad():
        for b in val:
            pred = ref(b["image"].unsqueeze(1))[0].argmax(1)[0].numpy()
            tot += eval_case(pred, b["label"][0].numpy(), 2)[0]
    assert abs(met["val_mean_dice"] - tot / 6) < 1e-4, (met["val_mean_dice"], tot / 6)
    assert abs(met["val_1_dice"] - tot / 6) < 1e-4


@pytest.mark.parametrize("amp_dtype", ["fp16", "bf16"])
def test_amp_gradscaler_matches_reference_state_machine(amp_dtype):
    """`--amp 1` (a19): autocast = 16-bit storage (fp16 like the reference's torch.cuda.amp, or bf16), GradScaler's state
    machine held on the device.  Loss scaling by a power of two is exact in bf16 (fp32's exponent range), so a clean AMP
    run reproduces the plain bf16 run; in fp16 it only changes which tiny gradients survive, so the runs agree to fp16
    accuracy.  An inf/NaN gradient skips the optimizer step (parameters, moments and step count untouched), halves the
    scale and clears the flag."""
    from fedicra_amd import ops
    from fedicra_amd.amp import GradScaler
    from fedicra_amd.flower_common import MyModel
    from fedicra_amd.flower_pCE_2D import MyClient
    from fedicra_amd.networks.unet import UNet
    from helpers import loader
    batches = loader(2, 4, 64, cid=0)
    finals, losses = [], []
    for amp in (False, True):
        for use_graph in ((False,) if not amp else (False, True)):
            args = _args(use_graph=use_graph, iters=4, amp=int(amp), amp_dtype=amp_dtype)
            ops.manual_seed(1)
            net = _mk(UNet, 1, 2, dtype=amp_dtype)
            client = MyClient(args, MyModel(args, net, batches, batches), batches, batches, amp=amp)
            assert net.compute_dtype() == {"fp16": torch.float16, "bf16": torch.bfloat16}[amp_dtype]
            client._train({"iter_global": 4, "iters": 4, "eval_iters": 8, "batch_size": 4, "stage": "fit"})
            finals.append(net.flat_state.clone())
            losses.append(list(client.last_losses))
            if amp:
                assert client.scaler.get_scale() == 65536.0 and int(client.scaler._tracker.item()) == 4
                # last_losses holds the UNscaled loss like the reference's logging
    # (fp16: AdamW's first steps are sign-like, m / sqrt(v), so a tiny gradient that survives in one run only moves its
    #  parameter by a full lr step; by the fourth iteration the two trajectories are a few 1e-2 apart for some dropout masks)
    assert np.allclose(losses[0][:3], losses[1][:3], rtol=0, atol=2e-3 if amp_dtype == "bf16" else 6e-3), (losses[0], losses[1])
    assert np.allclose(losses[0], losses[1], rtol=0, atol=2e-3 if amp_dtype == "bf16" else 3e-2), (losses[0], losses[1])
    assert abs(losses[0][0] - losses[1][0]) < 1e-6                       # the forward pass does not see the scale
    if amp_dtype == "bf16":
        assert np.allclose(losses[0][:2], losses[1][:2], atol=1e-5)
    assert np.allclose(losses[1], losses[2], atol=5e-3)
    # ---- skipped step
    from fedicra_amd.optim import FusedAdamW
    net = _mk(UNet, 1, 2, dtype=amp_dtype).train()
    opt = FusedAdamW(net, lr=0.01, base_lr=0.01, max_iterations=100)
    sc = GradScaler(init_scale=1024.0, growth_interval=2)
    b = batches[0]
    x, y = b["image"].unsqueeze(1).to(DEV), b["label"].to(DEV)

    def one_step(poison):
        ops.begin_iteration(x.device)
        opt.zero_grad()
        loss = ops.ce_loss(net(x)[0].permute(0, 2, 3, 1), y, 2)
        sc.scale(loss).backward()
        if poison:
            ops.flush_wgrad()
            net.flat_grads[5] = float("inf")
        sc.step(opt)
        sc.update()

    one_step(False)
    assert sc.get_scale() == 1024.0 and int(opt.steps[0].item()) == 1
    before = (net.flat_params.clone(), opt.m.clone(), opt.v.clone())
    one_step(True)
    assert torch.equal(net.flat_params, before[0]) and torch.equal(opt.m, before[1]) and torch.equal(opt.v, before[2])
    assert int(opt.steps[0].item()) == 1 and sc.get_scale() == 512.0 and float(sc._found_inf.item()) == 0.0
    one_step(False)
    one_step(False)                                       # two clean steps = growth_interval -> scale doubles
    assert int(opt.steps[0].item()) == 3 and sc.get_scale() == 1024.0
    assert not torch.equal(net.flat_params, before[0])


def test_fp16_autocast_overflow_drives_the_scaler():
    """fp16 for real: a loss scale far too large pushes the scaled gradients out of fp16's range; the kernels then store
    inf, the unscale pass finds it, the step is skipped and the scale halved -- until the gradients fit and training
    proceeds (torch.cuda.amp's dynamic loss scaling; with bf16 storage the same run never skips)."""
    from fedicra_amd import ops
    from fedicra_amd.amp import GradScaler
    from fedicra_amd.networks.unet import UNet
    from fedicra_amd.optim import FusedAdamW
    from helpers import loader
    b = loader(1, 4, 64, cid=0)[0]
    x, y = b["image"].unsqueeze(1).to(DEV), b["label"].to(DEV)
    outcome = {}
    for dtype in ("fp16", "bf16"):
        ops.manual_seed(2)
        net = _mk(UNet, 1, 2, dtype=dtype).train()
        opt = FusedAdamW(net, lr=0.01, base_lr=0.01, max_iterations=100)
        sc = GradScaler(init_scale=2.0 ** 30, growth_interval=1000)
        first, steps, loss = net.flat_params.clone(), [], None
        for it in range(24):
            ops.begin_iteration(x.device)
            opt.zero_grad()
            loss = ops.ce_loss(net(x)[0].permute(0, 2, 3, 1), y, 2)
            sc.scale(loss).backward()
            sc.step(opt)
            sc.update()
            steps.append(int(opt.steps[0].item()))
        outcome[dtype] = (sc.get_scale(), steps, float(loss.detach()), not torch.equal(first, net.flat_params))
    scale, steps, loss, moved = outcome["fp16"]
    assert steps[0] == 0, "the first fp16 steps must be skipped at a 2^30 loss scale"
    skipped = 24 - steps[-1]
    assert 2 <= skipped <= 20 and scale == 2.0 ** 30 / 2 ** skipped, (scale, steps)
    assert steps[-1] > 0 and moved and np.isfinite(loss)
    assert all(b_ - a_ in (0, 1) for a_, b_ in zip(steps, steps[1:])) and steps[-1] - steps[-4] == 3, steps   # settled
    scale_b, steps_b, loss_b, moved_b = outcome["bf16"]
    assert steps_b[-1] == 24 and scale_b == 2.0 ** 30 and moved_b and np.isfinite(loss_b)


def test_single_site_trainer_sgd_matches_torch_loop():
    """Unet_pCE.train (a15 single-site variant): SGD(momentum 0.9, wd 1e-4), pCE, the trainer's one-step-behind poly
    LR -- against the same loop written with torch.optim.SGD on the CPU oracle network, dropout masks pinned."""
    from fedicra_amd import ops
    from fedicra_amd.Unet_pCE import train
    from fedicra_amd.networks.unet import UNet
    from oracle.losses_ref import pce_loss
    from oracle.unet_ref import RefUNet, seeded_state
    from helpers import loader
    batches = loader(2, 4, 64, cid=0)
    args = _args(base_lr=0.03, max_iterations=4, in_chns=1)
    # oracle loop
    ref = RefUNet(1, 2)
    seeded_state(ref, 2022)
    ref.train()
    opt = torch.optim.SGD(ref.parameters(), lr=0.03, momentum=0.9, weight_decay=0.0001)
    torch.manual_seed(3)
    ref_loss, it = [], 0
    for _ in range(3):
        for b in batches:
            if it >= 4:
                break
            loss = pce_loss(ref(b["image"].unsqueeze(1))[0], b["label"], 2)
            opt.zero_grad()
            loss.backward()
            opt.step()
            lr_ = 0.03 * (1.0 - it / 4) ** 0.9
            for g in opt.param_groups:
                g["lr"] = lr_
            it += 1
            ref_loss.append(loss.item())
    # HIP
    net = _mk(UNet, 1, 2)
    ops.set_dropout_mask_provider(_mask_provider())
    try:
        torch.manual_seed(3)
        out = train(args, trainloader=batches, valloader=None, model=net)
    finally:
        ops.set_dropout_mask_provider(None)
    print("ref", ref_loss, "hip", out["loss"])
    assert len(out["loss"]) == 4
    assert out["lr"] == [0.03 * (1.0 - i / 4) ** 0.9 for i in range(4)]
    errs = [abs(a - b) for a, b in zip(out["loss"], ref_loss)]
    assert errs[0] < 1e-5 and errs[1] < 5e-4 and max(errs) < 1e-2, errs
    w_ref = ref.state_dict()["decoder.out_conv.weight"]
    w_hip = net.state_dict()["decoder.out_conv.weight"].cpu()
    assert (w_hip - w_ref).abs().max().item() < 2e-3


def test_head_models_match_reference_golden(golden):
    """unet_head / unet_multihead against the reference's own UNet_Head / UNet_MultiHead (g13): state_dict
    keys, eval logits / decoder features / auxiliary maps within 1e-4, and every parameter gradient of
    CE(main) + sum CE(aux at its scale) in eval mode (running statistics, no dropout: deterministic on both sides)."""
    from fedicra_amd.networks.unet import UNet_Head, UNet_MultiHead
    from helpers import assert_ck
    g = golden("g13_heads.npz")
    x = torch.from_numpy(g["x"]).unsqueeze(1).to(DEV)
    y = torch.from_numpy(g["dense"].astype(np.int64)).to(DEV)
    for tag, cls, heads, seed in (("head", UNet_Head, 1, 2025), ("multihead", UNet_MultiHead, 3, 2026)):
        m = _mk(cls, 1, 2, seed=seed).eval()
        assert list(m.state_dict().keys()) == [str(k) for k in g[f"{tag}/keys"]]
        o = m(x)
        assert len(o) == 6 + heads
        assert (o[0].detach().cpu() - torch.from_numpy(g[f"{tag}/eval_logits"])).abs().max().item() < 1e-4
        for i in range(2, 6):
            assert_ck(o[i].detach().float().cpu(), g[f"{tag}/eval_de{i-1}_ck"], rtol=2e-5, atol=1e-5, what=f"{tag} de{i-1}")
        for i in range(heads):
            ref = torch.from_numpy(g[f"{tag}/eval_aux{i+1}"])
            assert o[6 + i].shape == ref.shape
            assert (o[6 + i].detach().float().cpu() - ref).abs().max().item() < 1e-4
        loss = torch.nn.functional.cross_entropy(o[0].float(), y)
        for i in range(heads):
            a = o[6 + i].float()
            step = y.shape[-1] // a.shape[-1]
            loss = loss + torch.nn.functional.cross_entropy(a, y[:, ::step, ::step])
        assert abs(loss.item() - float(g[f"{tag}/loss"])) < 2e-5
        loss.backward()
        from fedicra_amd import ops
        ops.flush_wgrad()
        for k, p in m.named_parameters():
            assert_ck(p.grad.float().cpu(), g[f"{tag}/grad/{k}"], rtol=5e-3, atol=1e-6, what=f"{tag} grad {k}")


def test_captured_steps_are_cached_per_batch_shape(monkeypatch):
    """drop_last=False (the reference's DataLoader default) ends an epoch with a short batch: its static buffers and its
    captured iteration live beside the full-size ones -- after the first pass over the loader no shape change captures or
    runs eagerly again, and the short batch trains (its loss is finite and moves)."""
    from fedicra_amd.flower_common import MyModel
    from fedicra_amd.flower_pCE_2D import MyClient
    from fedicra_amd.networks.unet import UNet
    from helpers import loader
    captures = []
    real = torch.cuda.CUDAGraph

    class Counting(real):
        def __new__(cls, *a, **k):
            captures.append(1)
            return super().__new__(cls, *a, **k)

    monkeypatch.setattr(torch.cuda, "CUDAGraph", Counting)
    full = loader(2, 4, 64, cid=0)
    short = loader(1, 2, 64, cid=1)
    batches = [full[0], full[1], short[0]]                    # 4, 4, 2 images: an epoch of 10 with batch size 4
    args = _args(use_graph=True, iters=9)
    net = _mk(UNet, 1, 2)
    client = MyClient(args, MyModel(args, net, batches, batches), batches, batches)
    client._train({"iter_global": 9, "iters": 9, "eval_iters": 10, "batch_size": 4, "stage": "fit"})
    # per shape: eager first use, capture at the second, replay from then on -> exactly two captures in three epochs
    assert len(captures) == 2, len(captures)
    assert len(client._shapes) == 2 and all(len(v[2]) == 1 for v in client._shapes.values())
    assert np.isfinite(client.last_losses).all() and len(client.last_losses) == 9
    n0 = len(captures)
    client._train({"iter_global": 12, "iters": 3, "eval_iters": 10, "batch_size": 4, "stage": "fit"})
    assert len(captures) == n0                                # a later round replays all of it
