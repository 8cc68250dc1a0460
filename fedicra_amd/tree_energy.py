"""Tree-energy losses of the reference (/root/reference/code/flower_common.py:646-689 `TreeEnergyLoss`, :692-753
`MScaleAddTreeEnergyLoss`, :756-818 `MScaleRecurveTreeEnergyLoss`, :636-643 `tv_loss`), same constructor / forward signatures
and return tuples, on the device-resident tree filter (fedicra_amd/utils/tree_filter.py).

Round 5: the elementwise glue around the tree kernels is HIP too (csrc/treeglue.hip) -- what a loss computes BEFORE its trees
(softmax of the logits, bilinear resize of the guidance maps with align_corners=False, nearest resize of the unlabeled-pixel
mask and its count) is ONE launch (`fi_tree_prep_fwd`), the masked L1 of up to three filtered maps with the division by the
pixel count is ONE launch (`fi_tree_masked_l1_fwd`), each with one launch for its gradients; tv_loss's two 3x3 poolings keep
torch's first-extremum tie rule.  What is left to ATen in these classes' forward and backward: the zero fills of the fp64
accumulators and tv_loss's scalar division (`FI_TREE_GLUE=0` = the torch expressions of rounds 1-4, kept as the in-process
reference of tests/test_round5_gpu.py)."""
from __future__ import annotations

import os

import torch
import torch.nn as nn
import torch.nn.functional as F
from torch.autograd import Function

from . import _lib as L
from .utils.tree_filter import MinimumSpanningTree, TreeFilter2D

_GLUE = os.environ.get("FI_TREE_GLUE", "1") != "0"


def _use_glue(t):
    return _GLUE and t.is_cuda


class _TreePrep(Function):
    """(prob, low resized, rois, count, *highs resized) = everything a tree-energy loss needs before its trees, one launch.
    Gradients flow to the logits and to the head maps (the reference resizes `low_feats` and the mask under no_grad)."""

    @staticmethod
    def forward(ctx, preds, low_feats, roi_src, *highs):
        N, C, H, W = preds.shape
        dev = preds.device
        preds = preds if preds.dtype == torch.float32 else preds.float()
        srcs = [t if t.dtype == torch.float32 else t.float() for t in (low_feats,) + tuple(highs)]
        prob = torch.empty((N, C, H, W), dtype=torch.float32, device=dev)
        outs = [torch.empty((N, s.shape[1], H, W), dtype=torch.float32, device=dev) for s in srcs]
        rois = torch.empty((N, 1, H, W), dtype=torch.float32, device=dev)
        count = torch.zeros(1, dtype=torch.float64, device=dev)
        roi_src = roi_src.contiguous()
        if roi_src.dtype == torch.bool:
            roi8 = roi_src.view(torch.uint8)                        # one byte per element, 0 / 1: reinterpreted, not converted
        else:
            roi8 = roi_src if roi_src.dtype == torch.uint8 else (roi_src != 0).view(torch.uint8)
        L.tree_prep_fwd(preds, prob, list(zip(srcs, outs)), roi8, rois, count, N, H, W)
        ctx.save_for_backward(prob)
        ctx.high_shapes = [tuple(h.shape) for h in highs]
        ctx.mark_non_differentiable(outs[0], rois, count)
        return (prob, outs[0], rois, count) + tuple(outs[1:])

    @staticmethod
    def backward(ctx, dprob, _dlow, _drois, _dcount, *dhighs):
        (prob,) = ctx.saved_tensors
        N, C, H, W = prob.shape
        dpreds = None
        if ctx.needs_input_grad[0] and dprob is not None:
            dpreds = torch.empty_like(prob)
            dprob = dprob.contiguous()
        maps, gsrc = [], []
        for k, (g, shp) in enumerate(zip(dhighs, ctx.high_shapes)):
            if g is None or not ctx.needs_input_grad[3 + k]:
                gsrc.append(None)
                continue
            o = torch.empty(shp, dtype=torch.float32, device=prob.device)
            maps.append((g.contiguous(), o))
            gsrc.append(o)
        if dpreds is not None or maps:
            L.tree_prep_bwd(prob, dprob if dpreds is not None else None, dpreds, maps, N, H, W)
        return (dpreds, None, None) + tuple(gsrc)


class _MaskedL1(Function):
    """weight * sum_k sum(rois * |prob - AS_k|) / max(N, 1)  (flower_common.py:682-686, 745-751, 810-814), one launch each way."""

    @staticmethod
    def forward(ctx, prob, rois, count, weight, *maps):
        maps = [m.contiguous() for m in maps]
        acc = torch.zeros(L.TREE_TERMS + 1, dtype=torch.float64, device=prob.device)
        loss = torch.empty(1, dtype=torch.float32, device=prob.device)
        L.tree_masked_l1_fwd(prob, maps, rois, count, float(weight), acc, loss)
        ctx.save_for_backward(prob, rois, count, *maps)
        ctx.weight = float(weight)
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        prob, rois, count, *maps = ctx.saved_tensors
        gs = g.reshape(1).to(torch.float32).contiguous()
        dprob = torch.empty_like(prob) if ctx.needs_input_grad[0] else None
        das = [torch.empty_like(m) if ctx.needs_input_grad[4 + k] else None for k, m in enumerate(maps)]
        L.tree_masked_l1_bwd(prob, maps, rois, count, ctx.weight, gs, dprob, das)
        return (dprob, None, None, None) + tuple(das)


def _prep(preds, low_feats, unlabeled_ROIs, highs):
    """-> prob, low, rois, N, size, [highs resized]: one HIP launch, or (FI_TREE_GLUE=0 / CPU tensors) the torch expressions."""
    # (the kernel reads the mask as 0 / 1: a float or integer mask with other values -- which the reference would use as per-pixel
    #  weights, flower_common.py:776-778 -- takes the torch expressions)
    if _use_glue(preds) and unlabeled_ROIs.dtype == torch.bool:
        out = _TreePrep.apply(preds, low_feats, unlabeled_ROIs, *highs)
        return out[0], out[1], out[2], out[3], tuple(preds.shape[2:]), list(out[4:])
    preds = preds.float()
    with torch.no_grad():
        _, _, h, w = preds.size()
        low = F.interpolate(low_feats.float(), size=(h, w), mode="bilinear", align_corners=False)
        rois = F.interpolate(unlabeled_ROIs.unsqueeze(1).float(), size=(h, w), mode="nearest")
        N = rois.sum()
    prob = torch.softmax(preds, dim=1)
    hs = [F.interpolate(hf.float(), size=(h, w), mode="bilinear", align_corners=False) for hf in highs]
    return prob, low, rois, N, (h, w), hs


def _l1(prob, rois, N, weight, maps):
    """weight * (sum of the masked L1 terms, in the reference's order of additions) / max(N, 1)."""
    if _use_glue(prob):
        return _MaskedL1.apply(prob, rois, N, weight, *maps)
    tree_loss = (rois * torch.abs(prob - maps[0])).sum()
    for m in maps[1:]:
        tree_loss = tree_loss + (rois * torch.abs(prob - m)).sum()
    tree_loss = tree_loss / N.clamp(min=1)            # `if N > 0: tree_loss /= N` without a host sync (N = 0 => sum = 0)
    return weight * tree_loss


class TreeEnergyLoss(nn.Module):
    def __init__(self):
        super().__init__()
        self.mst_layers = MinimumSpanningTree(TreeFilter2D.norm2_distance)
        self.tree_filter_layers = TreeFilter2D(groups=1, sigma=0.02)

    def forward(self, preds, low_feats, high_feats, unlabeled_ROIs, weight):
        prob, low_feats, rois, N, size, highs = _prep(preds, low_feats, unlabeled_ROIs, [] if high_feats is None else [high_feats])
        guides = [low_feats] + highs
        trees = self.mst_layers.forward_many(guides)         # independent of the filter chain: built side by side
        AS = self.tree_filter_layers(feature_in=prob, embed_in=low_feats, tree=trees[0])
        if highs:
            AS = self.tree_filter_layers(feature_in=AS, embed_in=highs[0], tree=trees[1], low_tree=False)
        return _l1(prob, rois, N, weight, [AS]), AS


class MScaleRecurveTreeEnergyLoss(nn.Module):
    def __init__(self):
        super().__init__()
        self.mst_layers = MinimumSpanningTree(TreeFilter2D.norm2_distance)
        self.tree_filter_layers = TreeFilter2D(groups=1, sigma=0.02)

    def forward(self, preds, low_feats, high_feats_1, high_feats_2, high_feats_3, unlabeled_ROIs, weight):
        # the reference dereferences AS_1..AS_3 unconditionally at the end: all three maps are required
        prob, low_feats, rois, N, size, highs = _prep(preds, low_feats, unlabeled_ROIs, [high_feats_1, high_feats_2, high_feats_3])
        trees = self.mst_layers.forward_many([low_feats] + highs)     # the 4 trees do not depend on the filter chain
        AS = self.tree_filter_layers(feature_in=prob, embed_in=low_feats, tree=trees[0])
        outs, cur = [], AS
        for hf, tree in zip(highs, trees[1:]):
            cur = self.tree_filter_layers(feature_in=cur, embed_in=hf, tree=tree, low_tree=False)
            outs.append(cur)
        return _l1(prob, rois, N, weight, [outs[2]]), outs[0], outs[1], outs[2]


class MScaleAddTreeEnergyLoss(nn.Module):
    """/root/reference/code/flower_common.py:692-753: the PARALLEL multi-scale form -- every high-level tree filters the
    low-level affinity map `AS` itself (not the previous scale's output, as the recurve form does) and the three masked L1
    terms are ADDED before the division by the number of unlabeled pixels.  No reference script constructs it; built for the
    surface with the reference's signature and return tuple (weight * loss, AS_1, AS_2, AS_3)."""

    def __init__(self):
        super().__init__()
        self.mst_layers = MinimumSpanningTree(TreeFilter2D.norm2_distance)
        self.tree_filter_layers = TreeFilter2D(groups=1, sigma=0.02)

    def forward(self, preds, low_feats, high_feats_1, high_feats_2, high_feats_3, unlabeled_ROIs, weight):
        # as in the recurve form, the reference reads AS_1..AS_3 unconditionally: all three guidance maps are required
        prob, low_feats, rois, N, size, highs = _prep(preds, low_feats, unlabeled_ROIs, [high_feats_1, high_feats_2, high_feats_3])
        trees = self.mst_layers.forward_many([low_feats] + highs)
        AS = self.tree_filter_layers(feature_in=prob, embed_in=low_feats, tree=trees[0])
        outs = [self.tree_filter_layers(feature_in=AS, embed_in=hf, tree=tree, low_tree=False)
                for hf, tree in zip(highs, trees[1:])]
        return _l1(prob, rois, N, weight, outs), outs[0], outs[1], outs[2]      # the three terms, in the reference's order


class _TVLoss(Function):
    @staticmethod
    def forward(ctx, p):
        p = (p if p.dtype == torch.float32 else p.float()).contiguous()
        er = torch.empty_like(p)
        idx_e = torch.empty(p.shape, dtype=torch.uint8, device=p.device)
        idx_d, pos = torch.empty_like(idx_e), torch.empty_like(idx_e)
        acc = torch.zeros(1, dtype=torch.float64, device=p.device)
        L.tv_loss_fwd(p, er, idx_e, idx_d, pos, acc)
        ctx.save_for_backward(idx_e, idx_d, pos)
        ctx.shape = tuple(p.shape)
        return (acc / float(p.numel())).to(torch.float32)[0]

    @staticmethod
    def backward(ctx, g):
        idx_e, idx_d, pos = ctx.saved_tensors
        scratch = torch.empty(ctx.shape, dtype=torch.float32, device=idx_e.device)
        dp = torch.empty_like(scratch)
        L.tv_loss_bwd(idx_e, idx_d, pos, g.reshape(1).to(torch.float32).contiguous(), ctx.shape, scratch, dp)
        return dp


def tv_loss(predication):
    """/root/reference/code/flower_common.py:636-643: mean contour length of a soft prediction -- a 3x3 erosion (min-pool as
    the negated max-pool of the negation) followed by a 3x3 dilation minus the eroded map, rectified, averaged.  (Argument
    name as spelled in the reference.)  On the device: csrc/treeglue.hip, torch's tie rule kept; elsewhere the torch ops."""
    if _use_glue(predication) and predication.dim() == 4:
        return _TVLoss.apply(predication)
    eroded = -F.max_pool2d(-predication, (3, 3), 1, 1)
    contour = torch.relu(F.max_pool2d(eroded, (3, 3), 1, 1) - eroded)
    return torch.mean(torch.abs(contour))
