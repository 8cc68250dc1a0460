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


class medpy_binary:
    """``medpy.metric.binary`` (medpy==0.4.0, fed39v2.yaml:88; call sites /root/reference/code/val_2D.py:13-19) restated
    from the published 0.4.0 definitions -- third-party, absent from /root/reference and from this image: PARITY
    UNPINNED for these seven formulas.  Everything ABOVE them (the class rule ``== 1`` / ``>= 1``, the empty-prediction
    rule, the per-dataset and per-class folds of val_2D.py:9-22,66-74 and flower_common.py:122-136) is pinned by golden g8:
    the reference's own functions run with this namespace standing in for the missing module."""

    @staticmethod
    def _b(result, reference):
        return np.atleast_1d(np.asarray(result).astype(bool)), np.atleast_1d(np.asarray(reference).astype(bool))

    @staticmethod
    def dc(result, reference):
        r, g = medpy_binary._b(result, reference)
        inter = np.count_nonzero(r & g)
        try:
            return 2.0 * inter / float(np.count_nonzero(r) + np.count_nonzero(g))
        except ZeroDivisionError:
            return 0.0

    @staticmethod
    def jc(result, reference):
        r, g = medpy_binary._b(result, reference)
        return float(np.count_nonzero(r & g)) / float(np.count_nonzero(r | g))

    @staticmethod
    def precision(result, reference):
        r, g = medpy_binary._b(result, reference)
        tp, fp = np.count_nonzero(r & g), np.count_nonzero(r & ~g)
        try:
            return tp / float(tp + fp)
        except ZeroDivisionError:
            return 0.0

    @staticmethod
    def recall(result, reference):
        r, g = medpy_binary._b(result, reference)
        tp, fn = np.count_nonzero(r & g), np.count_nonzero(~r & g)
        try:
            return tp / float(tp + fn)
        except ZeroDivisionError:
            return 0.0

    @staticmethod
    def specificity(result, reference):
        r, g = medpy_binary._b(result, reference)
        tn, fp = np.count_nonzero(~r & ~g), np.count_nonzero(r & ~g)
        try:
            return tn / float(tn + fp)
        except ZeroDivisionError:
            return 0.0

    @staticmethod
    def ravd(result, reference):
        r, g = medpy_binary._b(result, reference)
        v1, v2 = np.count_nonzero(r), np.count_nonzero(g)
        if v2 == 0:
            raise RuntimeError("The second supplied array does not contain any binary object.")
        return (v1 - v2) / float(v2)

    @staticmethod
    def hd95(result, reference, voxelspacing=None, connectivity=1):
        r, g = medpy_binary._b(result, reference)
        if not r.any():
            raise RuntimeError("The first supplied array does not contain any binary object.")
        if not g.any():
            raise RuntimeError("The second supplied array does not contain any binary object.")
        return hd95_percase(r, g)


VAL_METRICS = ["dice", "hd95", "recall", "precision", "jc", "specificity", "ravd"]     # flower_common.py:121


def metric_percase(pred: np.ndarray, gt: np.ndarray) -> tuple:
    """val_2D.py:9-22: all seven metrics of one binary (prediction, ground truth) pair; zeros for an empty prediction."""
    pred, gt = np.asarray(pred) > 0, np.asarray(gt) > 0
    if pred.sum() > 0:
        b = medpy_binary
        return (b.dc(pred, gt), b.hd95(pred, gt), b.recall(pred, gt), b.precision(pred, gt), b.jc(pred, gt),
                b.specificity(pred, gt), b.ravd(pred, gt))
    return (0, 0, 0, 0, 0, 0, 0)


def eval_case_all(pred_lbl: np.ndarray, gt_lbl: np.ndarray, classes: int) -> list:
    """val_2D.py:66-74 with all seven columns: class 1 is ``== 1``, every class >= 2 is ``>= 1``."""
    return [metric_percase(pred_lbl == 1, gt_lbl == 1) if i == 1 else metric_percase(pred_lbl >= 1, gt_lbl >= 1)
            for i in range(1, classes)]


def evaluate_labels(preds, gts, classes: int, n_dataset: int = None) -> dict:
    """flower_common.py:122-136 on arg-max label maps: per-image metric rows summed, divided by len(dataset); per-class
    and class-mean entries under the reference's keys."""
    total = 0.0
    for p, g in zip(preds, gts):
        total = total + np.array(eval_case_all(p, g, classes), dtype=np.float64)
    total = total / (len(preds) if n_dataset is None else n_dataset)
    out = {}
    for ci in range(classes - 1):
        for mi, name in enumerate(VAL_METRICS):
            out["val_{}_{}".format(ci + 1, name)] = total[ci, mi]
    for mi, name in enumerate(VAL_METRICS):
        out["val_mean_{}".format(name)] = np.mean(total, axis=0)[mi]
    return out
