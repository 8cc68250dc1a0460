"""Generate tests/golden/*.npz from the REFERENCE ITSELF (run in the build container only).

    python -m oracle.gen_golden

Imports the reference's own modules from /root/reference/code through
oracle/_refshim.py (CPU, ``.cuda()`` no-op) and records inputs + outputs as small
fixtures.  Weights are NOT stored: both sides rebuild them with
oracle.unet_ref.seeded_state(seed).  The fixtures are data; no reference source
text is written anywhere.  /root/reference does not exist on the GPU box -- only
the committed .npz files travel.  torch version used: see meta['torch'].
"""
from __future__ import annotations

import argparse
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from oracle import _refshim  # noqa: E402

_refshim.install()

from oracle.unet_ref import seeded_state  # noqa: E402
from fedicra_amd.synth import phantom_batch  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")


def checksum(t) -> np.ndarray:
    """[sum, abs-sum, L2, 16 samples at fixed strided positions] as float64."""
    a = np.asarray(t.detach().cpu().numpy() if torch.is_tensor(t) else t, dtype=np.float64).ravel()
    idx = np.linspace(0, a.size - 1, 16).astype(np.int64)
    return np.concatenate([[a.sum(), np.abs(a).sum(), np.sqrt((a * a).sum())], a[idx]])


def state_checksums(model, prefix=""):
    return {prefix + k: checksum(v.double()) for k, v in model.state_dict().items()}


def ref_pcs_extra(model):
    return {f"encoder.pcs_list.{i}.{k}": v for i, p in enumerate(model.encoder.pcs_list)
            for k, v in p.state_dict().items()}


def save(name, **arrs):
    arrs["meta_torch"] = np.array(torch.__version__)
    np.savez_compressed(os.path.join(OUT, name), **arrs)
    print("wrote", name, {k: getattr(v, "shape", None) for k, v in list(arrs.items())[:6]}, "...")


def g2_unet():
    from networks.unet import UNet
    img, weak, dense = phantom_batch(4, 64, 1, 2, cid=0)
    x = torch.from_numpy(img).unsqueeze(1)
    m = UNet(1, 2)
    seeded_state(m, 2022)
    m.eval()
    with torch.no_grad():
        o = m(x)
    d = {"x": img, "eval_logits": o[0].numpy()}
    for i, f in enumerate(o[1]):
        d[f"eval_feat{i}_ck"] = checksum(f)
    for i in range(2, 6):
        d[f"eval_de{i-1}_ck"] = checksum(o[i])
    m.train()
    torch.manual_seed(7)
    o = m(x)
    d["train_logits_seed7"] = o[0].detach().numpy()
    for k, v in state_checksums(m, "after_train_fwd/").items():
        if "running" in k or "num_batches" in k:
            d[k] = v
    # 3-channel / 3-class variant (ODOC-like), eval only
    img3, _, _ = phantom_batch(2, 64, 3, 3, cid=1)
    m3 = UNet(3, 3)
    seeded_state(m3, 2023)
    m3.eval()
    with torch.no_grad():
        d["x3"] = img3
        d["eval_logits3"] = m3(torch.from_numpy(img3))[0].numpy()
    save("g2_unet_fwd.npz", **d)


def g2_unet_lc():
    from networks.unet import UNet_LC, UNet_LC_MultiHead
    img, _, _ = phantom_batch(4, 64, 1, 2, cid=3)
    x = torch.from_numpy(img).unsqueeze(1)
    m = UNet_LC(1, 2, 1, 8, 8, 3)
    seeded_state(m, 2022, extra=ref_pcs_extra(m))
    d = {"x": img}
    m.eval()
    with torch.no_grad():
        for e in (None, 0, 5):
            o = m(x, e)
            d[f"eval_logits_e{e}"] = o[0].numpy()
            d[f"eval_hmap_e{e}"] = o[6][-1].numpy()
            d[f"eval_aux_e{e}"] = o[7].numpy()
            d[f"eval_feat4_ck_e{e}"] = checksum(o[1][4])
    m.train()
    torch.manual_seed(11)
    o = m(x)
    d["train_logits_seed11"] = o[0].detach().numpy()
    d["train_hmap_seed11"] = o[6][-1].detach().numpy()
    d["train_aux_seed11"] = o[7].detach().numpy()
    mh = UNet_LC_MultiHead(1, 2, 1, 8, 8, 2)
    seeded_state(mh, 2024, extra=ref_pcs_extra(mh))
    mh.eval()
    with torch.no_grad():
        o = mh(x)
    d["mh_eval_logits"] = o[0].numpy()
    for i in (7, 8, 9):
        d[f"mh_eval_aux{i-6}_ck"] = checksum(o[i])
    save("g2_unet_lc_fwd.npz", **d)


def g3_losses():
    from utils.losses import pDLoss, DiceLoss
    rng = np.random.default_rng(33)
    d = {}
    for C in (2, 3):
        logits = torch.tensor(rng.normal(0, 2, (4, C, 64, 64)).astype(np.float32), requires_grad=True)
        lab = rng.integers(0, C, (4, 64, 64)).astype(np.uint8)
        lab[rng.random(lab.shape) < 0.9] = C
        lab_t = torch.from_numpy(lab)
        ce = torch.nn.CrossEntropyLoss(ignore_index=C)(logits, lab_t.long())
        (g_ce,) = torch.autograd.grad(ce, logits)
        probs = torch.softmax(logits.detach(), 1).requires_grad_(True)
        pd = pDLoss(C, ignore_index=C)(probs, lab_t.unsqueeze(1))
        (g_pd,) = torch.autograd.grad(pd, probs)
        dense = torch.from_numpy(rng.integers(0, C, (4, 1, 64, 64)).astype(np.uint8))
        probs2 = torch.softmax(logits.detach(), 1).requires_grad_(True)
        dl = DiceLoss(C)(probs2, dense)
        (g_dl,) = torch.autograd.grad(dl, probs2)
        d.update({f"logits{C}": logits.detach().numpy(), f"labels{C}": lab, f"dense{C}": dense.numpy(),
                  f"ce{C}": np.float64(ce.item()), f"ce_grad{C}": g_ce.numpy(),
                  f"pdice{C}": np.float64(pd.item()), f"pdice_grad{C}": g_pd.numpy(),
                  f"dice{C}": np.float64(dl.item()), f"dice_grad{C}": g_dl.numpy()})
    # all-ignored edge case: CE is nan (0/0) in torch
    logits = torch.zeros(1, 2, 8, 8)
    lab = torch.full((1, 8, 8), 2, dtype=torch.long)
    d["ce_all_ignored_isnan"] = np.array(bool(torch.isnan(torch.nn.CrossEntropyLoss(ignore_index=2)(logits, lab))))
    save("g3_losses.npz", **d)


def _args(**kw):
    a = argparse.Namespace(strategy="FedAvg", amp=0, model="unet", cid=0, min_num_clients=1, num_classes=2,
                           img_class="faz", base_lr=0.01, max_iterations=30000, iters=5, rep_iters=3, alpha=0.5,
                           snapshot_path="/tmp")
    a.__dict__.update(kw)
    return a


def _loader(n_batches, B, S, cid, in_chns=1, ncls=2):
    out = []
    for i in range(n_batches):
        img, weak, _ = phantom_batch(B, S, in_chns, ncls, cid=cid, index=i, labeled_frac=0.1)
        out.append({"image": torch.from_numpy(img), "label": torch.from_numpy(weak)})
    return out


def g4_train():
    """The reference's own MyClient._train (flower_pCE_2D.py:51-181), FedAvg, 'unet'."""
    import flower_pCE_2D as ref
    import flower_common as fc
    logs = []
    ref.log = lambda lvl, msg, *a: logs.append(msg)
    args = _args()
    loader = _loader(3, 4, 64, cid=0)
    net = ref.net_factory(args, net_type="unet", in_chns=1, class_num=2)
    seeded_state(net, 2022)
    model = fc.MyModel(args, net, loader, loader)
    client = ref.MyClient(args, model, loader, loader)
    torch.manual_seed(2022)
    cfg = {"iter_global": 5, "iters": 5, "eval_iters": 10, "batch_size": 4, "stage": "fit"}
    last, metrics = client._train(cfg)
    losses = [float(m.split("loss : ")[1].split(",")[0]) for m in logs if "loss :" in m]
    d = {"losses_6dp": np.array(losses), "last_loss": np.float64(last), "lr_after": np.float64(client.current_lr)}
    d.update(state_checksums(net, "state/"))
    d["out_conv_weight"] = net.state_dict()["decoder.out_conv.weight"].numpy().copy()
    d["in_conv0_weight"] = net.state_dict()["encoder.in_conv.conv_conv.0.weight"].numpy().copy()
    # second round: fresh AdamW, carried lr / current_iter (quirk 7)
    logs.clear()
    cfg["iters"] = 2
    args.iters = 2
    last2, _ = client._train(cfg)
    d["losses_round2_6dp"] = np.array([float(m.split("loss : ")[1].split(",")[0]) for m in logs if "loss :" in m])
    d["last_loss_round2"] = np.float64(last2)
    d.update(state_checksums(net, "state_r2/"))
    save("g4_train_unet.npz", **d)


def g5_fedicra_train():
    """Reference _train under FedICRA with 'unet_lc'.  As shipped, flower_pCE_2D.py:117-118
    unpacks the 8-element UNet_LC return into 7 names -> ValueError (recorded).  The vector is
    then produced with a wrapper *outside* the reference that truncates the return list to 7
    entries, which makes the reference's own lines 84-157 executable unchanged."""
    import flower_pCE_2D as ref
    import flower_common as fc
    logs = []
    ref.log = lambda lvl, msg, *a: logs.append(msg)
    K, cid = 3, 1
    args = _args(strategy="FedICRA", model="unet_lc", cid=cid, min_num_clients=K, iters=5, rep_iters=2, alpha=1.0)
    loader = _loader(2, 4, 64, cid=cid)
    net = ref.net_factory(args, net_type="unet_lc", in_chns=1, class_num=2)
    seeded_state(net, 2022, extra=ref_pcs_extra(net))

    class AsShipped(fc.MyModel):
        def forward(self, x, emb_idx=None):
            return self.model(x, emb_idx)

    class Repaired(fc.MyModel):
        def forward(self, x, emb_idx=None):
            return self.model(x, emb_idx)[:7]

    broken = False
    try:
        c0 = ref.MyClient(args, AsShipped(args, net, loader, loader), loader, loader)
        c0._train({"iter_global": 60, "iters": 5, "eval_iters": 10, "batch_size": 4, "stage": "fit"})
    except ValueError:
        broken = True
    seeded_state(net, 2022, extra=ref_pcs_extra(net))
    logs.clear()
    client = ref.MyClient(args, Repaired(args, net, loader, loader), loader, loader)
    torch.manual_seed(2022)
    last, metrics = client._train({"iter_global": 60, "iters": 5, "eval_iters": 10, "batch_size": 4, "stage": "fit"})
    d = {"as_shipped_raises_valueerror": np.array(broken),
         "losses_6dp": np.array([float(m.split("loss : ")[1].split(",")[0]) for m in logs if "loss :" in m]),
         "loss_ce_6dp": np.array([float(m.split("loss_ce: ")[1]) for m in logs if "loss :" in m]),
         "last_loss": np.float64(last), "loss_lc_last": np.float64(metrics[f"client_{cid}_loss_lc"])}
    d.update(state_checksums(net, "state/"))
    d["out_conv_weight"] = net.state_dict()["decoder.out_conv.weight"].numpy().copy()
    save("g5_fedicra_train.npz", **d)


def g7_ala():
    """The reference's own MyModel.set_weights, FedICRA branch (flower_common.py:494-624)."""
    import flower_common as fc
    K, cid = 3, 1
    args = _args(strategy="FedICRA", model="unet_lc", cid=cid, min_num_clients=K)
    loader = _loader(3, 4, 64, cid=cid)
    import flower_pCE_2D as ref
    d = {}
    for mode in ("eval", "train"):
        net = ref.net_factory(args, net_type="unet_lc", in_chns=1, class_num=2)
        seeded_state(net, 100, extra=ref_pcs_extra(net))            # "old local"
        donor = ref.net_factory(args, net_type="unet_lc", in_chns=1, class_num=2)
        seeded_state(donor, 200)                                    # "global"
        glob = [v.numpy().copy() for v in donor.state_dict().values()]
        model = fc.MyModel(args, net, loader, loader)
        getattr(model, mode)()
        # (a) iter_global <= 50 -> plain load
        model.set_weights(glob, {"iter_global": 50})
        d[f"{mode}/skip50_out_conv_ck"] = checksum(net.state_dict()["decoder.out_conv.weight"])
        assert model.start_phase is True
        # (b) first personalised call: converge loop
        seeded_state(net, 100, extra=ref_pcs_extra(net))
        torch.manual_seed(5)
        import io, contextlib
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            model.set_weights(glob, {"iter_global": 60})
        epochs1 = buf.getvalue().count("ALA epochs") - 1            # final line printed twice on convergence
        d[f"{mode}/first_epochs"] = np.array(epochs1)
        d.update(state_checksums(net, f"{mode}/first/"))
        d[f"{mode}/first_out_conv_weight"] = net.state_dict()["decoder.out_conv.weight"].numpy().copy()
        d[f"{mode}/first_mix_ck"] = np.stack([checksum(w) for w in model.fedaa_weights])
        # (c) second call with a different global: exactly one epoch
        seeded_state(donor, 300)
        glob2 = [v.numpy().copy() for v in donor.state_dict().values()]
        torch.manual_seed(6)
        with contextlib.redirect_stdout(io.StringIO()):
            model.set_weights(glob2, {"iter_global": 70})
        d.update(state_checksums(net, f"{mode}/second/"))
        d[f"{mode}/second_mix_ck"] = np.stack([checksum(w) for w in model.fedaa_weights])
        # (d) same global again -> early-out on sum(first param diff)==0: model == global entirely
        with contextlib.redirect_stdout(io.StringIO()):
            model.set_weights(glob2, {"iter_global": 80})
        d[f"{mode}/third_out_conv_ck"] = checksum(net.state_dict()["decoder.out_conv.weight"])
    save("g7_ala.npz", **d)


def g9_unet3d():
    """3D U-Net surface (a18): the reference's own unet_3D(1 -> 2 classes) on a 32^3 volume: eval logits, parameter
    gradients of a partial cross-entropy, and the state produced by its initialisation under a fixed torch seed."""
    from networks.unet_3D import unet_3D
    torch.manual_seed(11)
    m = unet_3D(n_classes=2, in_channels=1)
    d = {}
    d.update(state_checksums(m, "init_seed11/"))
    seeded_state(m, 2031)
    rng = np.random.default_rng(909)
    x = rng.random((1, 1, 32, 32, 32), dtype=np.float32)
    lab = rng.integers(0, 3, (1, 32, 32, 32)).astype(np.int64)          # 2 = ignore
    m.eval()                                                            # dropout off; InstanceNorm has no mode
    xt = torch.from_numpy(x)
    out = m(xt)
    d["x"], d["labels"], d["eval_logits"] = x, lab.astype(np.uint8), out.detach().numpy()
    loss = torch.nn.functional.cross_entropy(out, torch.from_numpy(lab), ignore_index=2)
    loss.backward()
    d["loss"] = np.array(loss.item())
    for k, p in m.named_parameters():
        d["grad_ck/" + k] = checksum(p.grad)
    d["grad/final.weight"] = m.final.weight.grad.numpy().copy()
    d["grad/conv1.conv1.0.weight"] = m.conv1.conv1[0].weight.grad.numpy().copy()
    d["grad/center.conv2.0.bias"] = m.center.conv2[0].bias.grad.numpy().copy()
    d["keys"] = np.array(list(m.state_dict().keys()))
    # a two-sample, 3-channel, 3-class input (different batch / channel plumbing); 32^3 is the smallest volume whose
    # centre level still has more than one voxel for InstanceNorm.  Inputs are regenerated from the seed by the tests.
    m3 = unet_3D(n_classes=3, in_channels=3)
    seeded_state(m3, 2032)
    m3.eval()
    x3 = np.random.default_rng(910).random((2, 3, 32, 32, 32), dtype=np.float32)
    with torch.no_grad():
        o3 = m3(torch.from_numpy(x3))
    d["eval_logits3_ck"] = checksum(o3)
    d["eval_logits3_block"] = o3[:, :, 10:14, 8:16, 8:16].numpy().copy()
    save("g9_unet3d.npz", **d)


def g10_gatedcrf():
    """Gated CRF loss (section 8f-2): the reference's own module, as its trainer calls it (Potts, no masks)."""
    from utils.gate_crf_loss import ModelLossSemsegGatedCRF
    rng = np.random.default_rng(77)
    d = {}
    cases = {"a": (2, 3, 24, 20, 3, [{"weight": 1, "xy": 6, "rgb": 0.1}], 5),
             "b": (1, 2, 17, 31, 1, [{"weight": 0.9, "xy": 6, "rgb": 0.1}, {"weight": 0.1, "xy": 6}], 3)}
    for name, (N, C, H, W, F_, desc, radius) in cases.items():
        logits = torch.from_numpy(rng.standard_normal((N, C, H, W)).astype(np.float32)).requires_grad_(True)
        sample = torch.from_numpy(rng.random((N, F_, H, W), dtype=np.float32))
        y = torch.softmax(logits, dim=1)
        y.retain_grad()
        loss = ModelLossSemsegGatedCRF()(y, desc, radius, sample.clone(), H, W)["loss"]
        loss.backward()
        d[f"{name}/logits"], d[f"{name}/sample"] = logits.detach().numpy(), sample.numpy()
        d[f"{name}/loss"], d[f"{name}/grad_y"] = np.array(loss.item()), y.grad.numpy()
        d[f"{name}/grad_logits"] = logits.grad.numpy()
    save("g10_gatedcrf.npz", **d)


def g11_augment():
    """Training-time augmentation (section 8f-3): the reference's RandomGenerator, seeded, on small images."""
    import random
    import dataloaders.dataset as D
    rng = np.random.default_rng(11)
    d = {}
    for name, cls_, shape in [("faz", "faz", (33, 33)), ("odoc", "odoc", (3, 24, 24)), ("polyp", "polyp", (3, 20, 20))]:
        n = 24
        imgs = rng.random((n,) + shape, dtype=np.float32)
        labs = rng.integers(0, D_NCLS[cls_] + 1, (n,) + shape[-2:]).astype(np.uint8)
        random.seed(100 + len(name))
        np.random.seed(200 + len(name))
        gen = D.RandomGenerator(list(shape[-2:]), img_class=cls_)
        outs = [gen({"image": imgs[i], "label": labs[i]}) for i in range(n)]
        d[f"{name}/image"], d[f"{name}/label"] = imgs, labs
        d[f"{name}/out_image"] = np.stack([o["image"].numpy() for o in outs])
        d[f"{name}/out_label"] = np.stack([o["label"].numpy() for o in outs])
        d[f"{name}/seeds"] = np.array([100 + len(name), 200 + len(name)])
    save("g11_augment.npz", **d)


def g12_vnet():
    """VNet (section 8-a18: Conv3d 3^3 / strided 2^3 / ConvTranspose3d 2^3, BatchNorm3d / GroupNorm(16) / InstanceNorm3d):
    the reference's own module, train mode without dropout -- logits, loss, gradient checksums."""
    from networks.vnet import VNet
    rng = np.random.default_rng(12)
    d = {}
    # batchnorm: 6 samples -- with 2, the 1^3 bottleneck normalises over two values per channel (+-1, decided by round-off)
    for nz, size, B in (("none", 16, 2), ("batchnorm", 16, 6), ("groupnorm", 16, 2), ("instancenorm", 32, 1)):
        torch.manual_seed(21)
        m = VNet(n_channels=1, n_classes=2, normalization=nz, has_dropout=False).train()
        x = torch.from_numpy(rng.random((B, 1, size, size, size), dtype=np.float32))
        y = torch.from_numpy(rng.integers(0, 2, (B, size, size, size))).long()
        out = m(x)
        loss = torch.nn.functional.cross_entropy(out, y)
        loss.backward()
        d[f"{nz}/x"], d[f"{nz}/y"] = x.numpy(), y.numpy().astype(np.uint8)
        d[f"{nz}/logits"], d[f"{nz}/loss"] = out.detach().numpy(), np.array(loss.item())
        for k, p_ in m.named_parameters():
            d[f"{nz}/grad/{k}"] = checksum(p_.grad)
        if nz == "batchnorm":
            d[f"{nz}/running_mean0"] = m.block_one.conv[1].running_mean.numpy().copy()
            d[f"{nz}/running_var0"] = m.block_one.conv[1].running_var.numpy().copy()
    save("g12_vnet.npz", **d)


def g13_heads():
    """The two head models without the LC encoder (section 8-a12: `unet_head`, `unet_multihead` -- the latter is what the
    tree-energy procedure trains under FedAvg): the reference's own UNet_Head / UNet_MultiHead, eval forward (logits,
    decoder features, auxiliary maps), a seeded train-mode forward (dropout RNG order) and the gradients of
    CE(main) + sum CE(aux at its own scale) in eval mode (no dropout, running statistics: deterministic on every side)."""
    from networks.unet import UNet_Head, UNet_MultiHead
    img, weak, dense = phantom_batch(3, 64, 1, 2, cid=5)
    x = torch.from_numpy(img).unsqueeze(1)
    d = {"x": img, "dense": dense.astype(np.uint8)}
    for tag, cls, seed, naux in (("head", UNet_Head, 2025, 1), ("multihead", UNet_MultiHead, 2026, 3)):
        m = cls(1, 2)
        seeded_state(m, seed)
        d[f"{tag}/keys"] = np.array(list(m.state_dict().keys()))
        m.eval()
        o = m(x)
        assert len(o) == 6 + naux
        d[f"{tag}/eval_logits"] = o[0].detach().numpy()
        for i in range(2, 6):
            d[f"{tag}/eval_de{i-1}_ck"] = checksum(o[i])
        for i in range(naux):
            d[f"{tag}/eval_aux{i+1}"] = o[6 + i].detach().numpy()
        y = torch.from_numpy(dense).long()
        loss = torch.nn.functional.cross_entropy(o[0], y)
        for i in range(naux):
            a = o[6 + i]
            step = y.shape[-1] // a.shape[-1]
            loss = loss + torch.nn.functional.cross_entropy(a, y[:, ::step, ::step])
        loss.backward()
        d[f"{tag}/loss"] = np.array(loss.item())
        for k, p_ in m.named_parameters():
            d[f"{tag}/grad/{k}"] = checksum(p_.grad)
        m.zero_grad()
        m.train()
        torch.manual_seed(13)
        o = m(x)
        d[f"{tag}/train_logits_seed13"] = o[0].detach().numpy()
        d[f"{tag}/train_aux1_seed13"] = o[6].detach().numpy()
    save("g13_heads.npz", **d)


def g14_metric_aggregation():
    """The strategy protocol's metric folds (section 8b-3): the reference's own get_evaluate_metrics_aggregation_fn /
    fit_metrics_aggregation_fn on random per-client metric dicts (3 clients, 3 classes, unequal example counts)."""
    import argparse
    import flower_common as rfc
    rng = np.random.default_rng(14)
    names = ["dice", "hd95", "recall", "precision", "jc", "specificity", "ravd"]
    args = argparse.Namespace(min_num_clients=3, num_classes=3)
    counts = [7, 19, 4]
    em, fm = [], []
    for c, n in enumerate(counts):
        m = {}
        for nm in names:
            for cls in (1, 2):
                m[f"client_{c}_val_{cls}_{nm}"] = float(rng.random())
            m[f"client_{c}_val_mean_{nm}"] = float(rng.random())
        em.append((n, m))
        fm.append((n, {f"client_{c}_total_loss": float(rng.random()), f"client_{c}_lr": float(rng.random())}))
    out = rfc.get_evaluate_metrics_aggregation_fn(args, names)(em)
    fout = rfc.fit_metrics_aggregation_fn(fm)
    d = {"counts": np.array(counts), "names": np.array(names)}
    for c, (n, m) in enumerate(em):
        d[f"in{c}_keys"], d[f"in{c}_vals"] = np.array(list(m.keys())), np.array(list(m.values()))
    for c, (n, m) in enumerate(fm):
        d[f"fit{c}_keys"], d[f"fit{c}_vals"] = np.array(list(m.keys())), np.array(list(m.values()))
    d["out_keys"], d["out_vals"] = np.array(list(out.keys())), np.array([float(v) for v in out.values()])
    d["fit_out_keys"], d["fit_out_vals"] = np.array(list(fout.keys())), np.array([float(v) for v in fout.values()])
    save("g14_metric_aggregation.npz", **d)


def g15_two_stream_sampler():
    """dataloaders/dataset.py:254-300 TwoStreamBatchSampler under np.random.seed: two epochs of index batches."""
    from dataloaders.dataset import TwoStreamBatchSampler
    d = {}
    for tag, (prim, sec, bs, sbs) in {"a": (list(range(23)), list(range(100, 107)), 6, 2),
                                      "b": (list(range(8)), list(range(50, 53)), 4, 3)}.items():
        np.random.seed(15)
        smp = TwoStreamBatchSampler(prim, sec, bs, sbs)
        ep = [np.array([list(map(int, b)) for b in smp]) for _ in range(2)]
        d[f"{tag}/args"] = np.array([len(prim), sec[0], len(sec), bs, sbs])
        d[f"{tag}/len"] = np.array(len(smp))
        d[f"{tag}/epoch0"], d[f"{tag}/epoch1"] = ep
    save("g15_two_stream_sampler.npz", **d)


def g16_base_datasets():
    """The reference's own BaseDataSets (dataset.py:63-183) over a dict-backed h5py on the release's directory layout:
    per client the sample lists (as sets: os.listdir order is the file system's), and per sample name the arrays
    __getitem__ returns for 'train' (label = sup_type) and 'val' (label = 'mask'), with and without the transform."""
    import random
    import tempfile
    import h5py
    from dataloaders.dataset import BaseDataSets, RandomGenerator
    from oracle.dataset_tree import dataset_tree
    store = {}

    class File(dict):
        def __init__(self, path, mode="r"):
            super().__init__(store[os.path.normpath(path)])
    h5py.File = File
    d = {}
    with tempfile.TemporaryDirectory() as root:
        dataset_tree(root, store)
        for client in ("client1", "client4", "client_all"):
            for split in ("train", "val"):
                ds = BaseDataSets(root, split, None, client, "scribble", "faz")
                names = sorted(ds.sample_list)
                d[f"{client}/{split}/names"] = np.array(names)
                by = {n: ds[i] for i, n in enumerate(ds.sample_list)}
                d[f"{client}/{split}/images"] = np.stack([by[n]["image"] for n in names])
                d[f"{client}/{split}/labels"] = np.stack([by[n]["label"] for n in names])
        ds = BaseDataSets(root, "train", RandomGenerator([16, 16], "faz"), "client2", "scribble", "faz")
        order = np.argsort(ds.sample_list)
        random.seed(16)
        np.random.seed(16)
        outs = [ds[int(i)] for i in order]
        d["aug/names"] = np.array([ds.sample_list[int(i)] for i in order])
        d["aug/images"] = np.stack([o["image"].numpy() for o in outs])
        d["aug/labels"] = np.stack([o["label"].numpy() for o in outs])
    save("g16_base_datasets.npz", **d)


def _install_tree_kernels():
    """The reference's Python glue (lib_tree_filter/modules/tree_filter.py, functions/*.py, the loss classes of
    flower_common.py:646-818) calls five entry points of its CUDA extension `tree_filter_cuda`, which cannot be built
    here.  For g17 ONLY, those five are served by the restated kernels of oracle/tree_ref.py (the MST by the reference's
    own boruvka.cpp compiled into oracle/_ref), so that g17 pins everything ABOVE the extension boundary -- edge index /
    weight construction, sorted gathers, exp(-d/sigma), the filter composition, ROI normalisation, autograd wiring --
    against the reference's own code.  The kernels below that boundary stay 'restated, parity unpinned' (DESIGN 5)."""
    import tree_filter_cuda as C
    from oracle import tree_ref as T
    assert T.have_reference_boruvka(), "build oracle/_ref first (python -c 'import __graft_entry__ as g; g.build()')"

    def mst_forward(edge_index, edge_weight, vertex_count):
        out = [T.mst_reference(edge_index[b].numpy(), edge_weight[b].numpy(), int(vertex_count))
               for b in range(edge_index.shape[0])]
        return torch.from_numpy(np.stack(out))

    def bfs_forward(edge_index, max_adj):
        si, sp, sc = [], [], []
        for b in range(edge_index.shape[0]):
            e = edge_index[b].numpy()
            width = int(np.abs(e[:, 1] - e[:, 0]).max())          # vertical grid edges join v and v + W
            a, p_, c, _ = T.bfs(e, e.shape[0] + 1, width)
            si.append(a), sp.append(p_), sc.append(c)
        return tuple(torch.from_numpy(np.stack(v)) for v in (si, sp, sc))

    def _orders(sidx, spar, schild, b):
        return sidx[b].numpy(), spar[b].numpy(), schild[b].numpy()

    def refine_forward(feat, w, sidx, spar, schild):
        r = [T.refine_forward(feat[b].numpy(), w[b].detach().numpy(), *_orders(sidx, spar, schild, b))
             for b in range(feat.shape[0])]
        return tuple(torch.from_numpy(np.stack([x[i] for x in r])) for i in range(5))

    def refine_backward_feature(feat, w, sidx, spar, schild, out, aggr, aggr_up, wsum, wsum_up, gout):
        return torch.from_numpy(np.stack([
            T.refine_backward_feature(gout[b].contiguous().numpy(), w[b].numpy(), *_orders(sidx, spar, schild, b),
                                      wsum[b].numpy()) for b in range(feat.shape[0])]))

    def refine_backward_weight(feat, w, sidx, spar, schild, out, aggr, aggr_up, wsum, wsum_up, gout):
        return torch.from_numpy(np.stack([
            T.refine_backward_weight(gout[b].contiguous().numpy(), w[b].numpy(), *_orders(sidx, spar, schild, b),
                                     out[b].numpy(), aggr[b].numpy(), aggr_up[b].numpy(), wsum[b].numpy(),
                                     wsum_up[b].numpy()) for b in range(feat.shape[0])]))
    C.mst_forward, C.bfs_forward, C.refine_forward = mst_forward, bfs_forward, refine_forward
    C.refine_backward_feature, C.refine_backward_weight = refine_backward_feature, refine_backward_weight


def g17_tree_glue():
    """TreeEnergyLoss (low-level tree only, and low + one high-level tree) and MScaleRecurveTreeEnergyLoss of the reference
    (flower_common.py:646-689, 756-818) over its own MinimumSpanningTree / TreeFilter2D modules, with the extension's
    five kernels served as _install_tree_kernels says: losses, filtered maps and the gradients w.r.t. the logits and the
    three head maps."""
    _install_tree_kernels()
    import flower_common as rfc
    rng = np.random.default_rng(17)
    B, S = 2, 16
    img = rng.random((B, 1, S, S), dtype=np.float32)
    roi = rng.random((B, S, S)) > 0.2
    mk = lambda *shape: (rng.standard_normal(shape) * 0.7).astype(np.float32)
    arrs = {"preds": mk(B, 2, S, S), "h1": mk(B, 2, S // 4, S // 4), "h2": mk(B, 2, S // 2, S // 2), "h3": mk(B, 2, S, S)}
    d = {"image": img, "roi": roi, **arrs}

    def leaves():
        return {k: torch.from_numpy(v.copy()).requires_grad_(True) for k, v in arrs.items()}
    low = torch.from_numpy(img).repeat(1, 3, 1, 1)
    unl = torch.from_numpy(roi)
    t = leaves()
    loss, AS = rfc.TreeEnergyLoss()(t["preds"], low, None, unl, 0.6)
    loss.backward()
    d["low/loss"], d["low/AS"], d["low/g_preds"] = np.array(loss.item()), AS.detach().numpy(), t["preds"].grad.numpy()
    t = leaves()
    loss, AS = rfc.TreeEnergyLoss()(t["preds"], low, t["h2"], unl, 0.6)
    loss.backward()
    d["high/loss"], d["high/AS"] = np.array(loss.item()), AS.detach().numpy()
    d["high/g_preds"], d["high/g_h2"] = t["preds"].grad.numpy(), t["h2"].grad.numpy()
    t = leaves()
    loss, a1, a2, a3 = rfc.MScaleRecurveTreeEnergyLoss()(t["preds"], low, t["h1"], t["h2"], t["h3"], unl, 0.6)
    loss.backward()
    d["ms/loss"] = np.array(loss.item())
    d["ms/AS1"], d["ms/AS2"], d["ms/AS3"] = a1.detach().numpy(), a2.detach().numpy(), a3.detach().numpy()
    for k in arrs:
        d[f"ms/g_{k}"] = t[k].grad.numpy()
    save("g17_tree_glue.npz", **d)


def g18_ours_train():
    """The reference's own MyClient._train of the README procedure (flower_pCE_2D_GateCRFMsacleTreeEnergyLoss_Ours.py:
    52-198: pCE + multi-scale tree energy + 0.1 x gated CRF), FedAvg, 'unet_multihead', 3 iterations at 32^2; the tree
    extension's kernels served as in g17.  Per-iteration loss / loss_ce / loss_tree from its log lines, final state."""
    _install_tree_kernels()
    import flower_pCE_2D_GateCRFMsacleTreeEnergyLoss_Ours as ref
    import flower_common as fc
    logs = []
    ref.log = lambda lvl, msg, *a: logs.append(msg)
    args = _args(model="unet_multihead", iters=3, tree_loss_weight=0.1, img_size=32)
    loader = _loader(2, 4, 32, cid=2)
    net = ref.net_factory(args, net_type="unet_multihead", in_chns=1, class_num=2)
    seeded_state(net, 2027)
    client = ref.MyClient(args, fc.MyModel(args, net, loader, loader), loader, loader)
    torch.manual_seed(2027)
    last, metrics = client._train({"iter_global": 3, "iters": 3, "eval_iters": 6, "batch_size": 4, "stage": "fit"})
    rows = [m for m in logs if "loss :" in m]
    d = {"losses_6dp": np.array([float(m.split("loss : ")[1].split(",")[0]) for m in rows]),
         "loss_ce_6dp": np.array([float(m.split("loss_ce: ")[1].split(",")[0]) for m in rows]),
         "loss_tree_6dp": np.array([float(m.split("loss_tree: ")[1].split(",")[0]) for m in rows]),
         "last_loss": np.float64(last), "lr_after": np.float64(client.current_lr)}
    d.update(state_checksums(net, "state/"))
    d["out_conv_weight"] = net.state_dict()["decoder.out_conv.weight"].numpy().copy()
    # the README's configuration: --model unet_lc_multihead --strategy FedICRA (freeze schedule + LC loss on top)
    logs.clear()
    K, cid = 3, 1
    args = _args(strategy="FedICRA", model="unet_lc_multihead", cid=cid, min_num_clients=K, iters=3, rep_iters=1, alpha=1.0,
                 tree_loss_weight=0.1, img_size=32)
    loader = _loader(2, 4, 32, cid=cid)
    net = ref.net_factory(args, net_type="unet_lc_multihead", in_chns=1, class_num=2)
    seeded_state(net, 2028, extra=ref_pcs_extra(net))
    class PassThrough(fc.MyModel):                  # MyModel.forward as shipped drops emb_idx (cf. g5); a wrapper OUTSIDE
        def forward(self, x, emb_idx=None):         # the reference hands it through so that :153-163 run unchanged
            return self.model(x, emb_idx)
    client = ref.MyClient(args, PassThrough(args, net, loader, loader), loader, loader)
    torch.manual_seed(2028)
    last, metrics = client._train({"iter_global": 60, "iters": 3, "eval_iters": 6, "batch_size": 4, "stage": "fit"})
    rows = [m for m in logs if "loss :" in m]
    d.update({"icra/losses_6dp": np.array([float(m.split("loss : ")[1].split(",")[0]) for m in rows]),
              "icra/loss_ce_6dp": np.array([float(m.split("loss_ce: ")[1].split(",")[0]) for m in rows]),
              "icra/loss_tree_6dp": np.array([float(m.split("loss_tree: ")[1].split(",")[0]) for m in rows]),
              "icra/last_loss": np.float64(last),
              "icra/loss_lc_last": np.float64(metrics.get(f"client_{cid}_loss_lc", np.nan))})
    d["icra/out_conv_weight"] = net.state_dict()["decoder.out_conv.weight"].numpy().copy()
    d["icra/running_mean0"] = net.state_dict()["encoder.in_conv.conv_conv.1.running_mean"].numpy().copy()
    save("g18_ours_train.npz", **d)


def g6_fedavg_counters():
    """The int64 -> float64 -> truncate path of a FedAvg round, through the reference's OWN wire functions
    (flower_common.py:488-489 ``MyModel.get_weights``, :627-633 ``set_weights`` of the non-personalised strategies) around
    the restated flwr ``aggregate`` (third-party, absent: oracle.fed_ref.fedavg_aggregate): K in {2, 5, 8} clients with the
    FAZ site sizes as n_k, every client a seeded UNet(1,2) whose BatchNorm counters were set to distinct values."""
    import flower_common as fc
    from networks.net_factory import net_factory
    from oracle import fed_ref
    n_all = [21, 13, 17, 59, 3, 21, 13, 17]
    d = {"n_all": np.array(n_all)}
    for K in (2, 5, 8):
        args = _args(strategy="FedAvg", min_num_clients=K)
        results = []
        for k in range(K):
            net = net_factory(args, net_type="unet", in_chns=1, class_num=2)
            seeded_state(net, 100 + k)
            j = 0
            for name, buf in net.named_buffers():
                if name.endswith("num_batches_tracked"):
                    buf.fill_(7 * k + 3 * j + 1)
                    j += 1
            w = fc.MyModel(args, net, None, None).get_weights({})
            results.append(([np.array(a) for a in w], n_all[k]))        # np.array: .numpy() aliases the parameters on CPU
        keys = list(net.state_dict().keys())
        agg = fed_ref.fedavg_aggregate(results)
        d[f"K{K}/dtypes"] = np.array([str(a.dtype) for a in agg])
        cidx = [i for i, k_ in enumerate(keys) if k_.endswith("num_batches_tracked")]
        d[f"K{K}/counters_f64"] = np.array([float(agg[i]) for i in cidx])
        recv = net_factory(args, net_type="unet", in_chns=1, class_num=2)
        seeded_state(recv, 7)
        fc.MyModel(args, recv, None, None).set_weights(agg, {"iter_global": 60})
        sd = recv.state_dict()
        d[f"K{K}/counters_loaded"] = np.array([int(sd[keys[i]]) for i in cidx], dtype=np.int64)
        for k_ in ("encoder.in_conv.conv_conv.0.weight", "encoder.down4.maxpool_conv.1.conv_conv.5.running_var",
                   "decoder.up1.conv.conv_conv.0.weight", "decoder.out_conv.weight", "decoder.out_conv.bias"):
            d[f"K{K}/ck/{k_}"] = checksum(sd[k_].double())
        d[f"K{K}/out_conv_weight"] = sd["decoder.out_conv.weight"].numpy().copy()
    d["keys"] = np.array(keys)
    save("g6_fedavg_counters.npz", **d)


def g8_eval_metrics():
    """The evaluation path above medpy: the reference's own ``calculate_metric_percase`` / ``test_single_volume``
    (val_2D.py:9-22,25-74) and ``evaluate`` (flower_common.py:122-136) with oracle.losses_ref.medpy_binary standing in for
    the absent ``medpy.metric.binary`` -- pins the class rule (class 1: ``== 1``; every class >= 2: ``>= 1``), the
    empty-prediction rule, the binarisation of non-binary inputs, the arg-max of the logits and the per-dataset /
    per-class folds.  A stateless stand-in network maps an image (its index is written into pixel [0,0]) to fixed
    logits.  Cases with an empty ground truth under a non-empty prediction are left out: medpy raises there and the
    reference does not catch it."""
    import types
    import flower_common as fc
    import val_2D
    from oracle import losses_ref
    binary = types.SimpleNamespace(**{k: getattr(losses_ref.medpy_binary, k)
                                      for k in ("dc", "hd95", "recall", "precision", "jc", "specificity", "ravd")})
    val_2D.metric = types.SimpleNamespace(binary=binary)
    rng = np.random.default_rng(8)
    S = 48
    yy, xx = np.mgrid[0:S, 0:S]

    def disk(cy, cx, r):
        return (yy - cy) ** 2 + (xx - cx) ** 2 <= r * r

    d = {}
    # --- direct calls, incl. non-binary integer inputs (the function binarises with > 0, in place)
    pairs = [(disk(20, 20, 9).astype(np.int64) * 3, disk(22, 21, 8).astype(np.int64) * 2),
             (np.zeros((S, S), np.int64), disk(22, 21, 8).astype(np.int64)),
             (disk(10, 10, 4) | disk(35, 30, 6), disk(12, 10, 5) | disk(35, 33, 5)),
             (disk(24, 24, 30), disk(24, 24, 12))]                        # prediction = the whole image
    d["direct/pred"] = np.stack([np.asarray(p).astype(np.int64) for p, _ in pairs])
    d["direct/gt"] = np.stack([np.asarray(g).astype(np.int64) for _, g in pairs])
    d["direct/out"] = np.array([[float(v) for v in val_2D.calculate_metric_percase(np.array(p), np.array(g))] for p, g in pairs])

    def make_set(ncls, specs):
        preds, gts = [], []
        for pred_disks, gt_disks in specs:
            p, g = np.zeros((S, S), np.int64), np.zeros((S, S), np.int64)
            for (cy, cx, r, c) in pred_disks:
                p[disk(cy, cx, r)] = c
            for (cy, cx, r, c) in gt_disks:
                g[disk(cy, cx, r)] = c
            preds.append(p)
            gts.append(g)
        logits = []
        for p in preds:
            l = rng.normal(0, 0.05, size=(ncls, S, S)).astype(np.float32)
            for c in range(ncls):
                l[c][p == c] += 4.0
            logits.append(l)
        return np.stack(preds), np.stack(gts), np.stack(logits)

    sets = {
        "faz": (2, 1, [([(20, 20, 9, 1)], [(22, 21, 8, 1)]),
                        ([], [(22, 21, 8, 1)]),                                           # empty prediction -> zeros
                        ([(10, 10, 4, 1), (35, 30, 6, 1)], [(12, 10, 5, 1), (35, 33, 5, 1)]),
                        ([(2, 2, 6, 1)], [(3, 3, 5, 1)]),                                # touching the image border
                        ([(24, 24, 3, 1)], [(24, 24, 14, 1)])]),
        "odoc": (3, 3, [([(24, 24, 14, 1), (24, 24, 6, 2)], [(25, 23, 13, 1), (25, 23, 5, 2)]),   # nested cup in disc
                         ([(24, 24, 10, 2)], [(24, 24, 12, 1), (24, 24, 5, 2)]),          # no class-1 pixel predicted
                         ([], [(24, 24, 12, 1), (24, 24, 5, 2)]),                         # nothing predicted
                         ([(20, 28, 9, 1)], [(22, 26, 10, 1), (22, 26, 4, 2)])]),         # class 2 never predicted
    }
    for name, (ncls, chns, specs) in sets.items():
        preds, gts, logits = make_set(ncls, specs)
        table = torch.from_numpy(logits)

        class Net:
            def eval(self):
                return self

            def __call__(self, x):
                idx = x.reshape(x.shape[0], -1)[:, 0].round().long()
                return [table[idx]]

        batches = []
        for i in range(len(preds)):
            img = rng.random((chns, S, S)).astype(np.float32)
            img[:, 0, 0] = i
            im = torch.from_numpy(img[0] if chns == 1 else img).unsqueeze(0)            # faz [1,H,W] / odoc [1,3,H,W]
            batches.append({"image": im, "label": torch.from_numpy(gts[i]).unsqueeze(0)})

        class Loader(list):
            pass

        loader = Loader(batches)
        loader.dataset = list(range(len(batches)))
        args = _args(num_classes=ncls, img_class=name)
        per_image = [val_2D.test_single_volume(b["image"], b["label"], Net(), classes=ncls) for b in batches]
        out = fc.evaluate(args, Net(), loader)
        d[f"{name}/pred"], d[f"{name}/gt"], d[f"{name}/logits"] = preds, gts, logits
        d[f"{name}/images"] = np.stack([b["image"].numpy()[0] for b in batches])
        d[f"{name}/per_image"] = np.array(per_image, dtype=np.float64)                  # [n, classes-1, 7]
        d[f"{name}/keys"] = np.array(list(out.keys()))
        d[f"{name}/vals"] = np.array([float(v) for v in out.values()])
    save("g8_eval_metrics.npz", **d)


def g19_minifed_dice(seeds=range(8), threads=(8, 1), rounds=12, iters=8, horizon=6):
    """Metric leg 3, "Dice vs CPU ref" (BASELINE.md section 3-5), from the reference's OWN classes: the miniature federation
    of fedicra_amd/minifed.py -- 2 FedAvg clients x `rounds` rounds x 8 local iterations -- driven through the reference's
    ``MyClient._train`` (flower_pCE_2D.py:51-181), ``MyModel.get_weights`` / ``set_weights`` (flower_common.py:488-489,
    627-633) around the restated flwr ``aggregate`` (absent third party) and its ``evaluate`` (flower_common.py:122-136,
    val_2D.py:9-74; medpy served by oracle.losses_ref.medpy_binary as in g8), for `seeds` x `threads` CPU threads: the
    spread of the reference's own final Dice under nothing but the dropout seed and the thread count is what a second
    correct implementation can be held to after ~100 AdamW steps per client.
    HORIZON part: ONE client, k = 1..`horizon` local iterations from the seeded state (mask seed 0), then the eval-mode
    logits of the first four validation cases and the validation Dice -- the vectors the HIP path must match to 1e-4 for
    as many steps as fp32 round-off allows (tests/test_round4_gpu.py asserts the measured horizon)."""
    import types
    import flower_pCE_2D as ref
    import flower_common as fc
    import val_2D
    from oracle import losses_ref, fed_ref
    from fedicra_amd.minifed import make_data
    binary = types.SimpleNamespace(**{k: getattr(losses_ref.medpy_binary, k)
                                      for k in ("dc", "hd95", "recall", "precision", "jc", "specificity", "ravd")})
    val_2D.metric = types.SimpleNamespace(binary=binary)
    ref.log = lambda *a, **k: None
    data, val = make_data()
    n_k = (3, 2)

    class Loader(list):
        pass

    vl = Loader(val)
    vl.dataset = list(range(len(val)))

    def clients(K, max_iterations):
        out = []
        for cid in range(K):
            args = _args(cid=cid, min_num_clients=K, iters=iters, max_iterations=max_iterations)
            net = ref.net_factory(args, net_type="unet", in_chns=1, class_num=2)
            seeded_state(net, 2022)
            model = fc.MyModel(args, net, data[cid], data[cid])
            out.append(ref.MyClient(args, model, data[cid], data[cid]))
        return out

    import contextlib, io
    d = {"seeds": np.array(list(seeds)), "threads": np.array(list(threads)), "rounds": np.int64(rounds), "iters": np.int64(iters)}
    dice = np.zeros((len(d["seeds"]), len(d["threads"])))
    last = np.zeros((len(d["seeds"]), len(d["threads"]), rounds * 2))
    have = os.path.join(OUT, "g19_minifed_dice.npz")
    reuse = os.path.exists(have) and os.environ.get("G19_SPREAD", "reuse") == "reuse"
    if reuse:          # the 16 federations take 4 minutes: G19_SPREAD=new re-runs them, otherwise only the horizon part is re-made
        old = np.load(have)
        reuse = (int(old["rounds"]) == rounds and list(old["seeds"]) == list(d["seeds"]) and list(old["threads"]) == list(d["threads"]))
        if reuse:
            dice, last = old["dice"], old["last_losses"]
    for ti, th in enumerate(() if reuse else threads):
        torch.set_num_threads(int(th))
        for si, seed in enumerate(seeds):
            cl = clients(2, 400)
            for rnd in range(rounds):
                res = []
                for cid in range(2):
                    torch.manual_seed(int(seed) + 100 * rnd + cid)
                    with contextlib.redirect_stdout(io.StringIO()):
                        l, _ = cl[cid]._train({"iter_global": rnd, "iters": iters, "eval_iters": 99, "batch_size": 4, "stage": "fit"})
                    last[si, ti, rnd * 2 + cid] = l
                    res.append((cl[cid].model.get_weights(None), n_k[cid]))
                glob = fed_ref.fedavg_aggregate(res)
                for c in cl:
                    c.model.set_weights(glob, {"iter_global": rnd})
            with contextlib.redirect_stdout(io.StringIO()):
                met = fc.evaluate(cl[0].args, cl[0].model.model, vl)
            dice[si, ti] = float(met["val_mean_dice"])
            print("g19", "threads", th, "seed", seed, "dice", dice[si, ti])
    d["dice"], d["last_losses"] = dice, last
    # horizon vectors: k = 0 is the seeded state itself (no step yet); threads 8 and 1 -- the reference against ITSELF under
    # nothing but another summation order is the yardstick for what any second fp32 implementation can share after k steps
    for th, sfx in ((8, ""), (1, "_t1")):
        torch.set_num_threads(th)
        hl, hd, hloss, ht = [], [], [], []
        for k in range(0, horizon + 1):
            c = clients(1, 200)[0]
            c.args.iters = k
            torch.manual_seed(0)
            l = 0.0
            with contextlib.redirect_stdout(io.StringIO()):
                if k:
                    l, _ = c._train({"iter_global": 0, "iters": k, "eval_iters": 99, "batch_size": 4, "stage": "fit"})
                met = fc.evaluate(c.args, c.model.model, vl)
            net = c.model.model.eval()
            with torch.no_grad():
                lg = torch.cat([net(b["image"].unsqueeze(1))[0] for b in val[:4]])
                # TRAIN-mode logits of the client's first batch under dropout seed 77 (batch statistics: what the training
                # iterations themselves see; the eval-mode ones above go through running statistics that k steps have barely
                # moved off their initial (0, 1), which blows round-off-level bias steps up by orders of magnitude)
                net.train()
                torch.manual_seed(77)
                lt = net(data[0][0]["image"].unsqueeze(1))[0]
            hl.append(lg.numpy().copy()); hd.append(float(met["val_mean_dice"])); hloss.append(float(l))
            ht.append(lt.numpy().copy())
        d["horizon_logits" + sfx], d["horizon_dice" + sfx] = np.stack(hl).astype(np.float32), np.array(hd)
        d["horizon_last_loss" + sfx], d["horizon_train_logits" + sfx] = np.array(hloss), np.stack(ht).astype(np.float32)
    torch.set_num_threads(8)
    save("g19_minifed_dice.npz", **d)


def g20_tree_add_tv():
    """MScaleAddTreeEnergyLoss and tv_loss of the reference (flower_common.py:692-753, 636-643; SURVEY 2.1 row 2b -- no
    reference script calls them), the tree extension's kernels served as in g17: loss, the three filtered maps, the
    gradients w.r.t. the logits and the three head maps; tv_loss value and gradient on a soft map."""
    _install_tree_kernels()
    import flower_common as rfc
    rng = np.random.default_rng(20)
    B, S = 2, 16
    img = rng.random((B, 1, S, S), dtype=np.float32)
    roi = rng.random((B, S, S)) > 0.2
    mk = lambda *shape: (rng.standard_normal(shape) * 0.7).astype(np.float32)
    arrs = {"preds": mk(B, 2, S, S), "h1": mk(B, 2, S // 4, S // 4), "h2": mk(B, 2, S // 2, S // 2), "h3": mk(B, 2, S, S)}
    d = {"image": img, "roi": roi, **arrs}
    t = {k: torch.from_numpy(v.copy()).requires_grad_(True) for k, v in arrs.items()}
    low = torch.from_numpy(img).repeat(1, 3, 1, 1)
    loss, a1, a2, a3 = rfc.MScaleAddTreeEnergyLoss()(t["preds"], low, t["h1"], t["h2"], t["h3"], torch.from_numpy(roi), 0.6)
    loss.backward()
    d["add/loss"] = np.array(loss.item())
    d["add/AS1"], d["add/AS2"], d["add/AS3"] = a1.detach().numpy(), a2.detach().numpy(), a3.detach().numpy()
    for k in arrs:
        d[f"add/g_{k}"] = t[k].grad.numpy()
    p = torch.softmax(torch.from_numpy(mk(B, 3, S, S)) * 3.0, dim=1).requires_grad_(True)
    tv = rfc.tv_loss(p)
    tv.backward()
    d["tv/p"], d["tv/loss"], d["tv/g"] = p.detach().numpy(), np.array(tv.item()), p.grad.numpy()
    save("g20_tree_add_tv.npz", **d)


D_NCLS = {"faz": 2, "odoc": 3, "polyp": 2}


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    torch.set_num_threads(8)
    which = sys.argv[1:] or ["g2_unet", "g2_unet_lc", "g3_losses", "g4_train", "g5_fedicra_train", "g6_fedavg_counters", "g7_ala", "g8_eval_metrics", "g9_unet3d", "g10_gatedcrf", "g11_augment", "g12_vnet", "g13_heads", "g14_metric_aggregation", "g15_two_stream_sampler", "g16_base_datasets", "g17_tree_glue", "g18_ours_train", "g19_minifed_dice", "g20_tree_add_tv"]
    for w in which:
        globals()[w]()
