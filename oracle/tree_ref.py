"""CPU restatement of the reference's tree-filter stack.  TEST INFRASTRUCTURE ONLY.

Follows /root/reference/code/utils/TreeEnergyLoss/kernels/lib_tree_filter/:
  modules/tree_filter.py:14-62   grid edge list + edge weights (squared L2 feature distance + 1) -> minimum spanning tree
  src/mst/boruvka.cpp:20-112     Boruvka, ties to the first edge in list order  (the reference's own file is compiled
                                 into oracle/_ref/libboruvka_ref.so by oracle/Makefile and used when present; the
                                 pure-Python Kruskal below with (weight, edge index) keys yields the same edge set)
  src/bfs/bfs.cu:19-98           adjacency + breadth-first order from vertex 0: sorted_index / sorted_parent / sorted_child
                                 (the reference's within-level order depends on atomics; here: frontier order, neighbours
                                 up, down, left, right -- any BFS order gives the same filter output)
  modules/tree_filter.py:92-110  tree edge weights exp(-d/sigma) (low-level tree) or exp(-d)
  src/refine/refine.cu:19-199    leaf->root aggregation, root->leaf propagation, edge-weight gradient recursion
  src/refine/refine.cu:201-370   refine_forward / refine_backward_feature / refine_backward_weight
  /root/reference/code/flower_common.py:646-689, 756-818   TreeEnergyLoss, MScaleRecurveTreeEnergyLoss

PARITY: the Boruvka step is pinned against the reference's own compiled source; the CUDA kernels (bfs, refine) cannot be
built here (THC headers, nvcc) -- they are restated from their source and checked for internal consistency (the analytic
gradients below are verified against finite differences in tests/test_oracle_golden.py): parity unpinned for those.
"""
from __future__ import annotations

import ctypes
import os

import numpy as np
import torch

_REF_SO = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_ref", "libboruvka_ref.so")


# ------------------------------------------------------------------------------------------- graph construction
def grid_index(H, W):
    """[E, 2] int32: vertical pairs ((h,w),(h+1,w)) row-major, then horizontal pairs ((h,w),(h,w+1))."""
    raw = np.arange(H * W, dtype=np.int32).reshape(H, W)
    row = np.stack([raw[:-1, :], raw[1:, :]], 2).reshape(-1, 2)
    col = np.stack([raw[:, :-1], raw[:, 1:]], 2).reshape(-1, 2)
    return np.concatenate([row, col], 0)


def grid_weights(fm: torch.Tensor) -> torch.Tensor:
    """[B, E] fp32: squared L2 feature distance + 1, torch ops as the reference applies them (tree_filter.py:27-34)."""
    B = fm.shape[0]
    d = lambda a, b: ((a - b) * (a - b)).sum(dim=1)
    wr = d(fm[:, :, :-1, :], fm[:, :, 1:, :]).reshape(B, -1)
    wc = d(fm[:, :, :, :-1], fm[:, :, :, 1:]).reshape(B, -1)
    return torch.cat([wr, wc], dim=1) + 1


def have_reference_boruvka():
    return os.path.exists(_REF_SO)


def mst_reference(index: np.ndarray, weight: np.ndarray, V: int) -> np.ndarray:
    """The reference's own boruvkaMST (compiled from its source): [V-1, 2] edges in its emission order."""
    lib = ctypes.CDLL(_REF_SO)
    idx = np.ascontiguousarray(index, dtype=np.int32)
    w = np.ascontiguousarray(weight, dtype=np.float32)
    out = np.zeros((V - 1, 2), dtype=np.int32)
    lib.ref_boruvka(ctypes.c_int(V), ctypes.c_int(len(w)), idx.ctypes.data_as(ctypes.c_void_p),
                    w.ctypes.data_as(ctypes.c_void_p), out.ctypes.data_as(ctypes.c_void_p))
    return out


def mst_kruskal(index: np.ndarray, weight: np.ndarray, V: int) -> np.ndarray:
    """Minimum spanning tree under the total order (weight, edge index): the tree Boruvka's first-seen tie rule yields.
    Returns the chosen edges in edge-index order, [V-1, 2]."""
    order = np.lexsort((np.arange(len(weight)), weight))
    parent = list(range(V))

    def find(a):
        while parent[a] != a:
            parent[a] = parent[parent[a]]
            a = parent[a]
        return a

    chosen = []
    for e in order:
        a, b = find(int(index[e, 0])), find(int(index[e, 1]))
        if a != b:
            parent[a] = b
            chosen.append(int(e))
    chosen.sort()
    return index[chosen].astype(np.int32)


def edge_set(edges: np.ndarray):
    return set((int(min(a, b)), int(max(a, b))) for a, b in edges)


def bfs(edges: np.ndarray, V: int, W: int):
    """sorted_index [V], sorted_parent [V] (positions), sorted_child [V, 4] (positions, 0 = none), level_start list."""
    adj = [[-1, -1, -1, -1] for _ in range(V)]          # up, down, left, right
    for a, b in edges:
        a, b = int(min(a, b)), int(max(a, b))
        if b == a + W:
            adj[a][1], adj[b][0] = b, a
        else:
            assert b == a + 1
            adj[a][3], adj[b][2] = b, a
    sidx, spar, vpar = [0], [0], [-1]
    schild = np.zeros((V, 4), dtype=np.int32)
    levels, lo = [0], 0
    while lo < len(sidx):
        hi = len(sidx)
        for i in range(lo, hi):
            k = 0
            for nb in adj[sidx[i]]:
                if nb >= 0 and nb != vpar[i]:
                    schild[i, k] = len(sidx)
                    k += 1
                    sidx.append(nb)
                    spar.append(i)
                    vpar.append(sidx[i])
        lo = hi
        levels.append(hi)
    assert len(sidx) == V, "edge list is not a spanning tree"
    if levels[-1] != V:
        levels.append(V)
    return np.array(sidx, np.int32), np.array(spar, np.int32), schild, levels


# ------------------------------------------------------------------------------------------- tree recursions
def aggr_up(x_orig, w, sidx, schild):
    """out[i] (sorted order) = x[sidx[i]] + sum_child out[child] * w[child]   (x None -> 1)."""
    V = len(sidx)
    out = np.zeros(V, dtype=np.float32)
    for i in range(V - 1, -1, -1):
        s = np.float32(1.0) if x_orig is None else np.float32(x_orig[sidx[i]])
        for c in schild[i]:
            if c <= 0:
                break
            s = np.float32(s + out[c] * w[c])
        out[i] = s
    return out


def prop_down(x_sorted, w, sidx, spar):
    """out[sidx[i]] (original order) = x[i] * (1 - w[i]^2) + out[sidx[par[i]]] * w[i], with w[0] = 0."""
    V = len(sidx)
    out = np.zeros(V, dtype=np.float32)
    for i in range(V):
        wi = np.float32(0.0) if i == 0 else np.float32(w[i])
        out[sidx[i]] = np.float32(x_sorted[i] * (np.float32(1.0) - wi * wi) + out[sidx[spar[i]]] * wi)
    return out


def tree_edge_weights(embed, sidx, spar, sigma, low_tree):
    """embed [C, V] -> w [V] (sorted order): exp(-|e_i - e_parent|^2 / sigma) (low) or exp(-|.|^2); w[root] = exp(0)."""
    src = embed[:, sidx]
    tar = src[:, spar]
    d = ((src - tar) * (src - tar)).sum(0)
    return np.exp(-d / np.float32(sigma)).astype(np.float32) if low_tree else np.exp(-d).astype(np.float32)


def refine_forward(feat, w, sidx, spar, schild):
    """feat [C, V] (original order) -> (out, aggr, aggr_up, wsum, wsum_up); w[0] is treated as 0 by the down pass."""
    C = feat.shape[0]
    up = np.stack([aggr_up(feat[c], w, sidx, schild) for c in range(C)])
    ag = np.stack([prop_down(up[c], w, sidx, spar) for c in range(C)])
    wup = aggr_up(None, w, sidx, schild)
    ws = prop_down(wup, w, sidx, spar)
    return ag / ws[None], ag, up, ws, wup


def refine_backward_feature(gout, w, sidx, spar, schild, wsum):
    C = gout.shape[0]
    gn = gout / wsum[None]
    return np.stack([prop_down(aggr_up(gn[c], w, sidx, schild), w, sidx, spar) for c in range(C)])


def _grad_rec(in_data, in_grad, out_data, w, sidx, spar):
    """refine.cu root_leaf_grad_kernel for one channel pair: in_data / in_grad in sorted order, out_data in original
    order; returns grad [V] (sorted) and leaves the propagated gradient in a copy of in_grad."""
    V = len(sidx)
    og = in_grad.copy()
    g = np.zeros(V, dtype=np.float32)
    for i in range(1, V):
        p = spar[i]
        wi = np.float32(w[i])
        left = in_grad[i] * (out_data[sidx[p]] - wi * in_data[i])
        right = in_data[i] * (og[p] - wi * in_grad[i])
        g[i] = np.float32(left + right)
        og[i] = np.float32(in_grad[i] * (np.float32(1.0) - wi * wi) + og[p] * wi)
    return g


def refine_backward_weight(gout, w, sidx, spar, schild, out, aggr, aggr_up_, wsum, wsum_up):
    """d loss / d w [V] (sorted order; 0 for the root)."""
    C = gout.shape[0]
    w0 = w.copy()
    w0[0] = 0.0
    gn = gout / wsum[None]
    fg = gn * out
    total = np.zeros(len(sidx), dtype=np.float32)
    for c in range(C):
        gn_up = aggr_up(gn[c], w0, sidx, schild)
        fg_up = aggr_up(fg[c], w0, sidx, schild)
        total += _grad_rec(aggr_up_[c], gn_up, aggr[c], w0, sidx, spar) - _grad_rec(wsum_up, fg_up, wsum, w0, sidx, spar)
    return total


# ------------------------------------------------------------------------------------------- torch glue (oracle modules)
class _RefineFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, feat, w, orders, low_tree):
        B = feat.shape[0]
        outs, saved = [], []
        for b in range(B):
            sidx, spar, schild, _ = orders[b]
            o, ag, up, ws, wup = refine_forward(feat[b].numpy(), w[b].numpy(), sidx, spar, schild)
            outs.append(o)
            saved.append((ag, up, ws, wup, o))
        ctx.saved, ctx.orders, ctx.low_tree = saved, orders, low_tree
        ctx.save_for_backward(w)
        return torch.from_numpy(np.stack(outs))

    @staticmethod
    def backward(ctx, gout):
        (w,) = ctx.saved_tensors
        gf, gw = [], []
        for b in range(gout.shape[0]):
            sidx, spar, schild, _ = ctx.orders[b]
            ag, up, ws, wup, o = ctx.saved[b]
            g = gout[b].contiguous().numpy()
            gf.append(refine_backward_feature(g, w[b].numpy(), sidx, spar, schild, ws))
            if not ctx.low_tree:
                gw.append(refine_backward_weight(g, w[b].numpy(), sidx, spar, schild, o, ag, up, ws, wup))
        gfeat = torch.from_numpy(np.stack(gf))
        return gfeat, (None if ctx.low_tree else torch.from_numpy(np.stack(gw))), None, None


def minimum_spanning_tree(guide: torch.Tensor, use_reference=True):
    """[B, V-1, 2] int32 tree edges for guide [B, C, H, W] (MinimumSpanningTree.forward, label=None)."""
    B, _, H, W = guide.shape
    idx = grid_index(H, W)
    wt = grid_weights(guide.detach()).numpy()
    f = mst_reference if (use_reference and have_reference_boruvka()) else mst_kruskal
    return np.stack([f(idx, wt[b], H * W) for b in range(B)])


def tree_filter(feature_in, embed_in, tree, sigma=0.02, low_tree=True):
    """TreeFilter2D.forward with groups = 1.  feature_in [B,C,H,W], embed_in [B,Ce,H,W] torch tensors (autograd ok)."""
    B, C, H, W = feature_in.shape
    V = H * W
    orders = [bfs(tree[b], V, W) for b in range(B)]
    sidx = torch.from_numpy(np.stack([o[0] for o in orders]).astype(np.int64))
    spar = torch.from_numpy(np.stack([o[1] for o in orders]).astype(np.int64))
    em = embed_in.reshape(B, embed_in.shape[1], V)
    src = torch.gather(em, 2, sidx.unsqueeze(1).expand(-1, em.shape[1], -1))
    tar = torch.gather(src, 2, spar.unsqueeze(1).expand(-1, em.shape[1], -1))
    d = ((src - tar) * (src - tar)).sum(dim=1)
    w = torch.exp(-d / sigma) if low_tree else torch.exp(-d)
    out = _RefineFn.apply(feature_in.reshape(B, C, V), w, orders, low_tree)
    return out.reshape(B, C, H, W)


def tree_energy_loss(preds, low_feats, unlabeled_rois, weight, sigma=0.02):
    """TreeEnergyLoss.forward restricted to the low-level tree (flower_common.py:646-689 with high_feats None path)."""
    import torch.nn.functional as F
    with torch.no_grad():
        B, _, h, w = preds.shape
        low = F.interpolate(low_feats, size=(h, w), mode="bilinear", align_corners=False)
        roi = F.interpolate(unlabeled_rois.unsqueeze(1).float(), size=(h, w), mode="nearest")
        N = roi.sum()
    prob = torch.softmax(preds, dim=1)
    tree = minimum_spanning_tree(low)
    AS = tree_filter(prob, low, tree, sigma, True)
    loss = (roi * torch.abs(prob - AS)).sum()
    if N > 0:
        loss = loss / N
    return weight * loss, AS


def mscale_recurve_tree_energy_loss(preds, low_feats, high1, high2, high3, unlabeled_rois, weight, sigma=0.02):
    """MScaleRecurveTreeEnergyLoss.forward (flower_common.py:756-818)."""
    import torch.nn.functional as F
    with torch.no_grad():
        B, _, h, w = preds.shape
        low = F.interpolate(low_feats, size=(h, w), mode="bilinear", align_corners=False)
        roi = F.interpolate(unlabeled_rois.unsqueeze(1).float(), size=(h, w), mode="nearest")
        N = roi.sum()
    prob = torch.softmax(preds, dim=1)
    AS = tree_filter(prob, low, minimum_spanning_tree(low), sigma, True)
    outs = []
    cur = AS
    for hf in (high1, high2, high3):
        hf = F.interpolate(hf, size=(h, w), mode="bilinear", align_corners=False)
        cur = tree_filter(cur, hf, minimum_spanning_tree(hf), sigma, False)
        outs.append(cur)
    loss = (roi * torch.abs(prob - outs[2])).sum()
    if N > 0:
        loss = loss / N
    return weight * loss, outs[0], outs[1], outs[2]


def mscale_add_tree_energy_loss(preds, low_feats, high1, high2, high3, unlabeled_rois, weight, sigma=0.02):
    """MScaleAddTreeEnergyLoss.forward (flower_common.py:692-753): each high-level tree filters the low-level map AS; the
    three masked L1 terms are summed, then divided by the number of unlabeled pixels."""
    import torch.nn.functional as F
    with torch.no_grad():
        B, _, h, w = preds.shape
        low = F.interpolate(low_feats, size=(h, w), mode="bilinear", align_corners=False)
        roi = F.interpolate(unlabeled_rois.unsqueeze(1).float(), size=(h, w), mode="nearest")
        N = roi.sum()
    prob = torch.softmax(preds, dim=1)
    AS = tree_filter(prob, low, minimum_spanning_tree(low), sigma, True)
    outs = []
    for hf in (high1, high2, high3):
        hf = F.interpolate(hf, size=(h, w), mode="bilinear", align_corners=False)
        outs.append(tree_filter(AS, hf, minimum_spanning_tree(hf), sigma, False))
    loss = (roi * torch.abs(prob - outs[0])).sum() + (roi * torch.abs(prob - outs[1])).sum()
    loss = loss + (roi * torch.abs(prob - outs[2])).sum()
    if N > 0:
        loss = loss / N
    return weight * loss, outs[0], outs[1], outs[2]


def tv_loss(p):
    """tv_loss (flower_common.py:636-643): 3x3 min-pool, then 3x3 max-pool minus it, rectified, mean."""
    import torch.nn.functional as F
    lo = -F.max_pool2d(-p, 3, 1, 1)
    return torch.relu(F.max_pool2d(lo, 3, 1, 1) - lo).abs().mean()
