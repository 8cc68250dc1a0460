"""CPU restatement of the hot-path losses and Dice metric (TEST INFRASTRUCTURE ONLY).

  * partial cross-entropy: torch ``CrossEntropyLoss(ignore_index=num_classes)``
    as constructed at /root/reference/code/flower_pCE_2D.py:57 and used at :124.
  * ``pdice_loss``  : /root/reference/code/utils/losses.py:195-232 (pDLoss).
  * ``dice_loss``   : /root/reference/code/utils/losses.py:156-192 (DiceLoss).
  * ``dice_percase``/``eval_case``: /root/reference/code/val_2D.py:9-22, 66-74.
    ``medpy.metric.binary.dc`` (medpy==0.4.0, fed39v2.yaml:88) is NOT vendored in
    /root/reference and not installed here: restated from its published
    definition  dc = 2*|P & G| / (|P| + |G|), 0.0 when both are empty
    -> "parity unpinned" for that one formula (the reference holds no test for it).
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F


def pce_loss(logits: torch.Tensor, labels: torch.Tensor, num_classes: int) -> torch.Tensor:
    """mean over non-ignored pixels of -log_softmax(logits)[label]; label==num_classes is 'unlabeled'."""
    return F.cross_entropy(logits, labels.long(), ignore_index=num_classes)


def _one_hot(target: torch.Tensor, n: int) -> torch.Tensor:
    # losses.py:162-168 / 201-207: class planes concatenated on dim 1
    return torch.cat([(target == i).float() for i in range(n)], dim=1)


def pdice_loss(probs: torch.Tensor, target: torch.Tensor, n_classes: int, ignore_index: int) -> torch.Tensor:
    """probs [B,C,H,W] (softmax already applied), target [B,1,H,W]."""
    # losses.py:221-222.  Reference quirk: the mask keeps its [B,1,H,W] shape while the
    # per-class score/target planes are [B,H,W], so ``score * target * mask`` broadcasts to
    # [B,B,H,W]: every image's ignore mask is applied to every image's product.  Restated
    # literally (sum_{b,b'} s[b'] t[b'] m[b]); for B == 1 it is the plain masked Dice.
    mask = (target != ignore_index).to(probs.dtype)         # [B,1,H,W]
    oh = _one_hot(target, n_classes)
    loss = 0.0
    for i in range(n_classes):                              # losses.py:209-218
        s, t = probs[:, i], oh[:, i]                        # [B,H,W]
        inter = torch.sum(s * t * mask)
        den = torch.sum(s * s * mask) + torch.sum(t * t * mask)
        loss = loss + (1 - (2 * inter + 1e-5) / (den + 1e-5))
    return loss / n_classes


def dice_loss(probs: torch.Tensor, target: torch.Tensor, n_classes: int) -> torch.Tensor:
    oh = _one_hot(target, n_classes)
    loss = 0.0
    for i in range(n_classes):                              # losses.py:170-178
        s, t = probs[:, i], oh[:, i]
        loss = loss + (1 - (2 * torch.sum(s * t) + 1e-5) / (torch.sum(s * s) + torch.sum(t * t) + 1e-5))
    return loss / n_classes


def dice_percase(pred: np.ndarray, gt: np.ndarray) -> float:
    """val_2D.py:9-22 restricted to the Dice column; medpy dc restated."""
    pred = pred > 0
    gt = gt > 0
    if pred.sum() == 0:                                     # val_2D.py:12,21-22
        return 0.0
    inter = np.count_nonzero(pred & gt)
    size = np.count_nonzero(pred) + np.count_nonzero(gt)
    return 2.0 * inter / float(size)


def eval_case(pred_lbl: np.ndarray, gt_lbl: np.ndarray, classes: int) -> list:
    """val_2D.py:66-74: class 1 is ``==1``, classes >=2 are ``>=1`` (cup/disc nesting)."""
    out = []
    for i in range(1, classes):
        if i == 1:
            out.append(dice_percase(pred_lbl == 1, gt_lbl == 1))
        else:
            out.append(dice_percase(pred_lbl >= 1, gt_lbl >= 1))
    return out


def hd95_percase(pred: np.ndarray, gt: np.ndarray) -> float:
    """medpy.metric.binary.hd95(result, reference) with its defaults (voxelspacing None, connectivity 1), restated from
    medpy 0.4.0's published algorithm (medpy is not installed here: PARITY UNPINNED for this formula; call site
    /root/reference/code/val_2D.py:14):
        border(m) = m XOR binary_erosion(m, generate_binary_structure(ndim, 1))
        sds(a, b) = distance_transform_edt(~border(b))[border(a)]
        hd95      = numpy.percentile(hstack(sds(result, reference), sds(reference, result)), 95)
    Returns NaN where medpy raises (an empty mask)."""
    from scipy.ndimage import binary_erosion, distance_transform_edt, generate_binary_structure
    pred, gt = pred > 0, gt > 0
    if not pred.any() or not gt.any():
        return float("nan")
    fp = generate_binary_structure(pred.ndim, 1)
    pb = pred ^ binary_erosion(pred, structure=fp, iterations=1)
    gb = gt ^ binary_erosion(gt, structure=fp, iterations=1)
    d1 = distance_transform_edt(~gb)[pb]
    d2 = distance_transform_edt(~pb)[gb]
    return float(np.percentile(np.hstack((d1, d2)), 95))
