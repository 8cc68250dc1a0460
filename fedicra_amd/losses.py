"""Hot-path losses with the reference's class surface, MI355X-native underneath.

Mirrors /root/reference/code/utils/losses.py ``DiceLoss`` (:156-192) and ``pDLoss`` (:195-232) -- same
constructor arguments, same ``forward(inputs, target, weight=None[, softmax=False])`` -- and the
``CrossEntropyLoss(ignore_index=num_classes)`` the training procedure builds
(/root/reference/code/flower_pCE_2D.py:57).  Every reduction is a HIP kernel behind the C ABI
(fi_pdice_*, fi_ce_*).  pDLoss reproduces the reference's ``[B,1,H,W]`` mask broadcasting to
``[B,B,H,W]`` (oracle/losses_ref.py documents and pins it against the reference).
"""
from __future__ import annotations

import torch
import torch.nn as nn
from torch.autograd import Function

from . import _lib as L
from . import ops


def _nhwc_f32(x_nchw):
    v = x_nchw.permute(0, 2, 3, 1)
    if v.dtype != torch.float32:
        v = v.float()
    return v if v.is_contiguous() else v.contiguous()


class _PDice(Function):
    @staticmethod
    def forward(ctx, probs_nhwc, labels_u8, ignore_index, weight):
        C = probs_nhwc.shape[-1]
        acc = torch.zeros(3 * C, dtype=torch.float64, device=probs_nhwc.device)
        L.pdice_fwd(probs_nhwc, labels_u8, ignore_index, acc)
        loss = torch.empty(1, dtype=torch.float32, device=probs_nhwc.device)
        L.pdice_finalize(acc, weight, C, loss)
        ctx.save_for_backward(probs_nhwc, labels_u8, acc, weight)
        ctx.ignore = ignore_index
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        probs, labels, acc, weight = ctx.saved_tensors
        d = torch.empty_like(probs)
        L.pdice_bwd(probs, labels, ctx.ignore, acc, weight, g.reshape(1).float().contiguous(), d)
        return d, None, None, None


def _dice(inputs, target, n_classes, ignore_index, weight):
    assert inputs.dim() == 4 and inputs.shape[1] == n_classes, "predict & target shape do not match"
    t = target.reshape(target.shape[0], *target.shape[-2:]).to(torch.uint8).contiguous()
    w = None
    if weight is not None:
        w = torch.as_tensor(weight, dtype=torch.float32, device=inputs.device).contiguous()
    return _PDice.apply(_nhwc_f32(inputs), t, int(ignore_index), w)


class DiceLoss(nn.Module):
    """losses.py:156-192.  ``inputs`` are probabilities [B,C,H,W] unless ``softmax=True``; ``target`` [B,1,H,W]."""

    def __init__(self, n_classes):
        super().__init__()
        self.n_classes = n_classes

    def forward(self, inputs, target, weight=None, softmax=False):
        if softmax:
            inputs = torch.softmax(inputs, dim=1)
        return _dice(inputs, target, self.n_classes, -1, weight)


class pDLoss(nn.Module):
    """losses.py:195-232: Dice restricted to labeled pixels (``target != ignore_index``)."""

    def __init__(self, n_classes, ignore_index):
        super().__init__()
        self.n_classes = n_classes
        self.ignore_index = ignore_index

    def forward(self, inputs, target, weight=None):
        return _dice(inputs, target, self.n_classes, self.ignore_index, weight)


class CrossEntropyLoss(nn.Module):
    """``torch.nn.CrossEntropyLoss(ignore_index=...)`` on [B,C,H,W] logits / [B,H,W] labels (mean over labeled pixels)."""

    def __init__(self, ignore_index=-100):
        super().__init__()
        self.ignore_index = ignore_index

    def forward(self, logits, target):
        return ops.ce_loss(_nhwc_f32(logits), target, self.ignore_index)
