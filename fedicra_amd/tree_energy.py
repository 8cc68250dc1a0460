"""Tree-energy losses of the reference (/root/reference/code/flower_common.py:646-689 `TreeEnergyLoss`, :756-818
`MScaleRecurveTreeEnergyLoss`), same constructor / forward signatures and return tuples, on the device-resident tree
filter (fedicra_amd/utils/tree_filter.py).  The elementwise glue (softmax, bilinear resize of the guidance maps, masked
L1) stays on torch ops exactly as in the reference; every tree operation is a libfedicra_hip.so launch."""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F

from .utils.tree_filter import MinimumSpanningTree, TreeFilter2D


def _prep(preds, low_feats, unlabeled_ROIs):
    with torch.no_grad():
        _, _, h, w = preds.size()
        low_feats = F.interpolate(low_feats.float(), size=(h, w), mode="bilinear", align_corners=False)
        rois = F.interpolate(unlabeled_ROIs.unsqueeze(1).float(), size=(h, w), mode="nearest")
        N = rois.sum()
    return low_feats, rois, N, (h, w)


class TreeEnergyLoss(nn.Module):
    def __init__(self):
        super().__init__()
        self.mst_layers = MinimumSpanningTree(TreeFilter2D.norm2_distance)
        self.tree_filter_layers = TreeFilter2D(groups=1, sigma=0.02)

    def forward(self, preds, low_feats, high_feats, unlabeled_ROIs, weight):
        preds = preds.float()
        low_feats, rois, N, size = _prep(preds, low_feats, unlabeled_ROIs)
        prob = torch.softmax(preds, dim=1)
        guides = [low_feats]
        if high_feats is not None:
            high_feats = F.interpolate(high_feats.float(), size=size, mode="bilinear", align_corners=False)
            guides.append(high_feats)
        trees = self.mst_layers.forward_many(guides)         # independent of the filter chain: built side by side
        AS = self.tree_filter_layers(feature_in=prob, embed_in=low_feats, tree=trees[0])
        if high_feats is not None:
            AS = self.tree_filter_layers(feature_in=AS, embed_in=high_feats, tree=trees[1], low_tree=False)
        tree_loss = (rois * torch.abs(prob - AS)).sum()
        tree_loss = tree_loss / N.clamp(min=1)        # `if N > 0: tree_loss /= N` without a host sync (N = 0 => sum = 0)
        return weight * tree_loss, AS


class MScaleRecurveTreeEnergyLoss(nn.Module):
    def __init__(self):
        super().__init__()
        self.mst_layers = MinimumSpanningTree(TreeFilter2D.norm2_distance)
        self.tree_filter_layers = TreeFilter2D(groups=1, sigma=0.02)

    def forward(self, preds, low_feats, high_feats_1, high_feats_2, high_feats_3, unlabeled_ROIs, weight):
        preds = preds.float()
        low_feats, rois, N, size = _prep(preds, low_feats, unlabeled_ROIs)
        prob = torch.softmax(preds, dim=1)
        # the reference dereferences AS_1..AS_3 unconditionally at the end: all three maps are required
        highs = [F.interpolate(hf.float(), size=size, mode="bilinear", align_corners=False)
                 for hf in (high_feats_1, high_feats_2, high_feats_3)]
        trees = self.mst_layers.forward_many([low_feats] + highs)     # the 4 trees do not depend on the filter chain
        AS = self.tree_filter_layers(feature_in=prob, embed_in=low_feats, tree=trees[0])
        outs, cur = [], AS
        for hf, tree in zip(highs, trees[1:]):
            cur = self.tree_filter_layers(feature_in=cur, embed_in=hf, tree=tree, low_tree=False)
            outs.append(cur)
        tree_loss = (rois * torch.abs(prob - outs[2])).sum()
        tree_loss = tree_loss / N.clamp(min=1)        # `if N > 0: tree_loss /= N` without a host sync (N = 0 => sum = 0)
        return weight * tree_loss, outs[0], outs[1], outs[2]


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
        preds = preds.float()
        low_feats, rois, N, size = _prep(preds, low_feats, unlabeled_ROIs)
        prob = torch.softmax(preds, dim=1)
        # as in the recurve form, the reference reads AS_1..AS_3 unconditionally: all three guidance maps are required
        highs = [F.interpolate(hf.float(), size=size, mode="bilinear", align_corners=False)
                 for hf in (high_feats_1, high_feats_2, high_feats_3)]
        trees = self.mst_layers.forward_many([low_feats] + highs)
        AS = self.tree_filter_layers(feature_in=prob, embed_in=low_feats, tree=trees[0])
        outs = [self.tree_filter_layers(feature_in=AS, embed_in=hf, tree=tree, low_tree=False)
                for hf, tree in zip(highs, trees[1:])]
        tree_loss = (rois * torch.abs(prob - outs[0])).sum() + (rois * torch.abs(prob - outs[1])).sum()
        tree_loss = tree_loss + (rois * torch.abs(prob - outs[2])).sum()          # the reference's order of additions
        tree_loss = tree_loss / N.clamp(min=1)
        return weight * tree_loss, outs[0], outs[1], outs[2]


def tv_loss(predication):
    """/root/reference/code/flower_common.py:636-643: mean contour length of a soft prediction -- a 3x3 erosion (min-pool as
    the negated max-pool of the negation) followed by a 3x3 dilation minus the eroded map, rectified, averaged.  (Argument
    name as spelled in the reference.)  Elementwise / pooling glue on torch ops, differentiable through them."""
    eroded = -F.max_pool2d(-predication, (3, 3), 1, 1)
    contour = torch.relu(F.max_pool2d(eroded, (3, 3), 1, 1) - eroded)
    return torch.mean(torch.abs(contour))
