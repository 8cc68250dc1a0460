"""CPU restatement of the federated hot path (TEST INFRASTRUCTURE ONLY).

  * ``get_weights`` / ``set_weights_plain``: /root/reference/code/flower_common.py:488-489, 627-633.
  * ``set_weights_ala`` (FedICRA adaptive local aggregation): flower_common.py:494-624.
  * ``local_train`` (``MyClient._train``): /root/reference/code/flower_pCE_2D.py:51-181.
  * ``fedavg_aggregate``: third-party ``flwr==1.0.0`` (fed39v2.yaml:74)
    ``flwr.server.strategy.aggregate.aggregate`` -- NOT vendored in /root/reference,
    not installed here.  Restated from its published v1.0.0 definition
        total = sum(n_k);  w_i = reduce(np.add, [w_{k,i} * n_k for k]) / total
    (python-int n_k: fp32 arrays stay fp32, int64 arrays become float64 by the true
    divide).  "parity unpinned" for this one function -- reference call sites:
    flower_common.py:11-17, 262, 451.  The int64 -> float64 -> ``torch.tensor`` ->
    ``load_state_dict`` (copy_ truncates toward zero) path is reproduced in
    ``set_weights_plain``.

Known reference defects restated as *working* code (documented in LOG.md section 2):
  * flower_pCE_2D.py:117-118 unpacks UNet_LC's 8-element return list into 7 names
    (ValueError as shipped); the restatement indexes out[0] (logits) / out[6] (heat-maps).
"""
from __future__ import annotations

from collections import OrderedDict
from functools import reduce
from typing import List, Sequence, Tuple

import numpy as np
import torch
import torch.nn as nn

from .losses_ref import pce_loss

ALA_KEYS = ("out_conv", "up4", "up3", "up2", "up1")        # flower_common.py:506
OUT_CONV = ("decoder.out_conv.weight", "decoder.out_conv.bias")   # flower_pCE_2D.py:86


# ------------------------------------------------------------------ weights I/O
def get_weights(model: nn.Module) -> List[np.ndarray]:
    # .copy(): on CPU ``.numpy()`` aliases parameter storage
    return [v.detach().cpu().numpy().copy() for v in model.state_dict().values()]


def set_weights_plain(model: nn.Module, weights: Sequence[np.ndarray]) -> None:
    sd = OrderedDict((k, torch.tensor(v)) for k, v in zip(model.state_dict().keys(), weights))
    model.load_state_dict(sd, strict=False)


def fedavg_aggregate(results: Sequence[Tuple[Sequence[np.ndarray], int]]) -> List[np.ndarray]:
    total = sum(n for _, n in results)
    scaled = [[layer * n for layer in w] for w, n in results]
    return [reduce(np.add, layers) / total for layers in zip(*scaled)]


# ------------------------------------------------------------------ FedICRA ALA
def ala_local_names(model: nn.Module) -> List[str]:
    return [n for n, _ in model.named_parameters() if any(k in n for k in ALA_KEYS)]


def set_weights_ala(model: nn.Module, weights: Sequence[np.ndarray], batches: Sequence[dict], *,
                    num_classes: int, iter_global: int, start_phase: bool, img_class: str = "faz",
                    eta: float = 1.0, num_pre_loss: int = 10, threshold: float = 0.1,
                    max_epochs: int = 10_000):
    """Returns (start_phase_after, epoch_losses, mix_weights or None).

    ``batches``: re-iterable list of {'image','label'} (the trainloader).
    Variable naming follows the *content*, not the reference's swapped names
    (SURVEY.md section 0 item 4): ``old`` = local weights before loading,
    ``glob`` = freshly loaded global weights.
    """
    names = [n for n, _ in model.named_parameters()]
    old = {n: p.detach().clone() for n, p in model.named_parameters()}       # :498 deepcopy
    set_weights_plain(model, weights)                                         # :499-502
    glob = {n: p.detach().clone() for n, p in model.named_parameters()}

    if torch.sum(old[names[0]] - glob[names[0]]) == 0:                        # :520-522
        return start_phase, [], None
    if iter_global <= 50:                                                     # :524-526
        return start_phase, [], None

    local = ala_local_names(model)
    # temp model = copy of (global) model incl. buffers and train/eval mode (:503)
    import copy
    temp = copy.deepcopy(model)
    tparams = dict(temp.named_parameters())
    for n, p in tparams.items():                                              # :542-546
        p.requires_grad = n in local
    mix = {n: torch.ones_like(glob[n]) for n in local}                        # :549-550 (always re-init, quirk 3)
    with torch.no_grad():
        for n in local:                                                       # :553-555
            tparams[n].copy_(glob[n] + (old[n] - glob[n]) * mix[n])

    losses = []
    epochs = 0
    while True:
        loss = None
        for b in batches:                                                     # :566-602
            x, y = b["image"], b["label"]
            if img_class == "faz":
                x = x.unsqueeze(1)
            logits = temp(x)[0]
            loss = pce_loss(logits, y, num_classes)
            for p in tparams.values():
                p.grad = None
            loss.backward()
            # AdamW(lr=0).step() changes nothing (:560, 587)
            with torch.no_grad():
                for n in local:
                    g = tparams[n].grad
                    if g is None:
                        continue
                    mix[n] = torch.clamp(mix[n] - eta * (g * (old[n] - glob[n])), 0, 1)
                for n in local:
                    tparams[n].copy_(glob[n] + (old[n] - glob[n]) * mix[n])
        losses.append(float(loss.item()))
        epochs += 1
        if not start_phase:                                                   # :611-612
            break
        if len(losses) > num_pre_loss and np.std(losses[-num_pre_loss:]) < threshold:   # :615
            break
        if epochs >= max_epochs:
            break
    with torch.no_grad():                                                     # :623-624
        for n, p in model.named_parameters():
            if n in local:
                p.copy_(tparams[n])
    return False, losses, mix


# ------------------------------------------------------------------ local training
class TrainState:
    """Client-side counters carried across rounds (flower_common.py:45-47)."""

    def __init__(self, base_lr: float):
        self.current_iter = 0
        self.current_lr = base_lr
        self.sampled_batches: list = []


def local_train(model: nn.Module, state: TrainState, batches: Sequence[dict], *, iters: int,
                num_classes: int, base_lr: float, max_iterations: int, img_class: str = "faz",
                strategy: str = "FedAvg", rep_iters: int = 3, alpha: float = 0.5,
                cid: int = 0, num_clients: int = 1):
    """Returns (last_loss, metrics dict with per-iteration losses)."""
    model.train()                                                             # :52
    opt = torch.optim.AdamW(model.parameters(), lr=state.current_lr, betas=(0.9, 0.999),
                            eps=1e-8, weight_decay=1e-2, amsgrad=False)       # :55 (fresh every round)
    hist, hist_ce, hist_lc = [], [], []
    n_b = len(batches)
    for i_iter in range(iters):
        if state.current_iter % n_b == 0:                                     # :66-70
            state.sampled_batches = list(batches)
        b = state.sampled_batches[state.current_iter % n_b]
        x, y = b["image"], b["label"]
        if img_class == "faz":
            x = x.unsqueeze(1)
        if strategy == "FedICRA":                                             # :84-101
            head_phase = i_iter < iters - rep_iters
            for n, p in model.named_parameters():
                p.requires_grad = (n in OUT_CONV) == head_phase
        out = model(x)
        logits = out[0]
        loss_ce = pce_loss(logits, y, num_classes)                            # :124
        loss = loss_ce
        loss_lc = None
        if strategy == "FedICRA":                                             # :128-139
            hm = out[6][-1]
            acc = 0
            for other in range(num_clients):
                if other == cid:
                    continue
                with torch.no_grad():
                    hm_o = model(x, other)[6][-1]
                acc = acc + torch.nn.functional.mse_loss(hm, hm_o.detach())
            loss_lc = -acc / (num_clients - 1)
            loss = torch.add(loss, loss_lc, alpha=alpha)
        opt.zero_grad()
        loss.backward()
        opt.step()
        state.current_iter += 1
        hist.append(float(loss.item()))
        hist_ce.append(float(loss_ce.item()))
        if loss_lc is not None:
            hist_lc.append(float(loss_lc.item()))
        lr_ = base_lr * (1.0 - state.current_iter / max_iterations) ** 0.9    # :154
        for g in opt.param_groups:
            g["lr"] = lr_
        state.current_lr = lr_
    return hist[-1], {"loss": hist, "loss_ce": hist_ce, "loss_lc": hist_lc, "lr": state.current_lr}


def local_train_ours(model: nn.Module, state: TrainState, batches: Sequence[dict], *, iters: int, num_classes: int,
                     base_lr: float, max_iterations: int, tree_loss_weight: float, img_class: str = "faz",
                     strategy: str = "FedAvg", rep_iters: int = 3, alpha: float = 0.5, cid: int = 0, num_clients: int = 1,
                     lc_model: bool = False):
    """MyClient._train of the README procedure (flower_pCE_2D_GateCRFMsacleTreeEnergyLoss_Ours.py:52-198):
    loss = CE(ignore = num_classes) + MScaleRecurveTreeEnergyLoss(logits, image x3, aux1..3, label == num_classes, w)
           + 0.1 * GatedCRF(softmax(logits), [{weight 1, xy 6, rgb 0.1}], radius 5, image)   [+ alpha * loss_lc]
    with the pCE client's batch selection, freeze schedule, per-round AdamW and poly learning rate."""
    from .gatedcrf_ref import gated_crf_loss
    from .tree_ref import mscale_recurve_tree_energy_loss
    model.train()                                                             # :53
    opt = torch.optim.AdamW(model.parameters(), lr=state.current_lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2,
                            amsgrad=False)                                    # :56
    hist = {"loss": [], "loss_ce": [], "loss_tree": [], "loss_crf": [], "loss_lc": []}
    n_b = len(batches)
    for i_iter in range(iters):
        if state.current_iter % n_b == 0:                                     # :72-76
            state.sampled_batches = list(batches)
        b = state.sampled_batches[state.current_iter % n_b]
        x, y = b["image"], b["label"]
        if img_class == "faz":
            x = x.unsqueeze(1)
        if strategy == "FedICRA":                                             # :89-107
            head_phase = i_iter < iters - rep_iters
            for n, p in model.named_parameters():
                p.requires_grad = (n in OUT_CONV) == head_phase
        out = model(x)                                                        # :112-128
        logits = out[0]
        heatmaps, aux = (out[6], out[7:10]) if lc_model else (None, out[6:9])
        soft = torch.softmax(logits, dim=1)
        loss_ce = pce_loss(logits, y, num_classes)                            # :135
        unlabeled = (y == num_classes)                                        # :136
        three = x.repeat(1, 3, 1, 1) if img_class == "faz" else x             # :138-141
        loss_tree = mscale_recurve_tree_energy_loss(logits, three, aux[0], aux[1], aux[2], unlabeled, tree_loss_weight)[0]
        loss_crf = gated_crf_loss(soft, [{"weight": 1, "xy": 6, "rgb": 0.1}], 5, x)[0]       # :143-150
        loss = loss_ce + loss_tree + 0.1 * loss_crf                           # :151
        if strategy == "FedICRA":                                             # :153-163
            acc = 0
            for other in range(num_clients):
                if other == cid:
                    continue
                with torch.no_grad():
                    hm_o = model(x, other)[-4]
                acc = acc + torch.nn.functional.mse_loss(heatmaps[-1], hm_o[-1].detach())
            loss_lc = -acc / (num_clients - 1)
            loss = torch.add(loss, loss_lc, alpha=alpha)
            hist["loss_lc"].append(float(loss_lc.item()))
        opt.zero_grad()
        loss.backward()
        opt.step()
        state.current_iter += 1
        for k, v in (("loss", loss), ("loss_ce", loss_ce), ("loss_tree", loss_tree), ("loss_crf", loss_crf)):
            hist[k].append(float(v.item()))
        lr_ = base_lr * (1.0 - state.current_iter / max_iterations) ** 0.9    # :180
        for g in opt.param_groups:
            g["lr"] = lr_
        state.current_lr = lr_
    hist["lr"] = state.current_lr
    return hist["loss"][-1], hist
