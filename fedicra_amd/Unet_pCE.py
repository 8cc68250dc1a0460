"""Single-site (non-federated) trainer of the reference, on the HIP path.

Mirrors ``train(args, snapshot_path)`` of /root/reference/code/Unet_pCE.py:63-226: SGD(momentum 0.9, weight decay 1e-4)
(:88-89), partial cross-entropy with ``ignore_index = num_classes`` (:90,124), the poly learning rate computed from the
iteration count BEFORE it is incremented (:132-136 -- one step behind the federated client's schedule), the optional
``--amp`` GradScaler flow (:69-70,125-128) and, every 20 iterations, the validation pass whose loss is
``0.5 * (CE + pDLoss(softmax))`` averaged over the validation set plus the per-class Dice table (:159-200).

Data loading, TensorBoard and checkpoint files of the script are outside the hot-path scope: the
loaders are passed in, the scalars the script would log are returned.
"""
from __future__ import annotations

import numpy as np
import torch

from . import ops
from .amp import GradScaler
from .flower_common import evaluate
from .losses import pDLoss
from .networks.net_factory import net_factory
from .networks.unet import set_compute_dtype
from .optim import FusedSGD


def _batch(args, sampled_batch, dev):
    x, y = sampled_batch["image"], sampled_batch["label"]
    if args.img_class == "faz":
        x = x.unsqueeze(1)                                   # Unet_pCE.py:104-106
    return x.to(dev), y.to(dev)


def train(args, snapshot_path=None, trainloader=None, valloader=None, model=None, val_every=20):
    """Returns {"loss": [...], "lr": [...], "val": [(iter, loss_val, metrics dict), ...], "model": model}."""
    base_lr, num_classes, max_iterations = args.base_lr, args.num_classes, args.max_iterations
    amp = bool(getattr(args, "amp", 0))
    if model is None:
        # Unet_pCE.py:71 calls net_factory without `args` (a TypeError as shipped); the working call is this one
        model = net_factory(args, net_type=args.model, in_chns=args.in_chns, class_num=num_classes)
    dev = model.flat_params.device
    if amp:
        scaler = GradScaler()
        set_compute_dtype(model, getattr(args, "amp_dtype", "fp16"))
    model.train()
    opt = FusedSGD(model, lr=base_lr, base_lr=base_lr, max_iterations=max_iterations, momentum=0.9, weight_decay=1e-4)
    opt.iter.fill_(-1)          # advance_lr() increments first: starting at -1 gives lr = f(iter_num BEFORE increment)
    ops.set_dropout_seed_offset(opt.iter)
    dice_loss = pDLoss(num_classes, ignore_index=num_classes)
    n_b = len(trainloader)
    max_epoch = max_iterations // n_b + 1
    hist = []
    out = {"loss": [], "lr": [], "val": [], "model": model}
    iter_num = 0
    for _epoch in range(max_epoch):
        for sampled_batch in trainloader:
            x, y = _batch(args, sampled_batch, dev)
            ops.begin_iteration(dev)
            outputs = model(x)[0]
            loss = ops.ce_loss(outputs.permute(0, 2, 3, 1), y, num_classes)
            opt.zero_grad()
            if amp:
                scaler.scale(loss).backward()
                scaler.step(opt)
                scaler.update()
            else:
                loss.backward()
                opt.step()
            opt.advance_lr()
            out["lr"].append(base_lr * (1.0 - iter_num / max_iterations) ** 0.9)
            iter_num += 1
            hist.append(loss.detach())
            if iter_num > 0 and iter_num % val_every == 0 and valloader is not None:
                model.eval()
                loss_val = 0.0
                with torch.no_grad():
                    for vb in valloader:
                        xv, yv = _batch(args, vb, dev)
                        ov = model(xv)[0]
                        lce = ops.ce_loss(ov.permute(0, 2, 3, 1), yv, num_classes)
                        loss_val = loss_val + 0.5 * (lce + dice_loss(torch.softmax(ov, dim=1), yv.unsqueeze(1)))
                n_val = len(valloader.dataset) if hasattr(valloader, "dataset") else len(valloader)
                out["val"].append((iter_num, float(loss_val) / n_val, evaluate(args, model, valloader)))
                model.train()
            if iter_num >= max_iterations:
                break
        if iter_num >= max_iterations:
            break
    out["loss"] = torch.stack(hist).cpu().tolist() if hist else []
    return out
