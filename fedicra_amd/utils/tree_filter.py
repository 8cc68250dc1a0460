"""Tree filter with the reference's module surface (`MinimumSpanningTree`, `TreeFilter2D`:
/root/reference/code/utils/TreeEnergyLoss/kernels/lib_tree_filter/modules/tree_filter.py:9-150 and functions/{mst,bfs,
refine}.py), on the HIP kernels of csrc/tree.hip.  The reference's `tree_filter_cuda` extension round-trips the graph
through the host for the spanning tree and walks the tree with spin-waiting 64-thread workgroups; here the spanning tree,
the breadth-first order and the level-parallel recursions all stay on the device.

Supported: what the tree-energy losses use -- `MinimumSpanningTree(TreeFilter2D.norm2_distance)` without the `label`
argument, `TreeFilter2D(groups=1)` with the default distance.  Tensors are NCHW fp32 (the loss runs in fp32 like the
reference's extension: it only has `float*` kernels).
"""
from __future__ import annotations

import torch
from torch import nn
from torch.autograd import Function

from .. import _lib as L


class Tree:
    """What `MinimumSpanningTree.forward` returns: the edge list [B, V-1, 2] int32 (the reference's return value) plus the
    image size the BFS needs."""

    def __init__(self, edges, H, W):
        self.edges, self.H, self.W = edges, H, W

    @property
    def shape(self):
        return self.edges.shape


def _parallel(*branches):
    """Run independent launch sequences as parallel branches: branch 0 on the current stream, the others on side HIP
    streams forked from / joined to it (inside a stream capture they become parallel branches of the hipGraph).  A tree
    recursion is a chain of ~1500 dependent steps on a few dozen workgroups, so two of them side by side cost the time
    of one.  Outputs are allocated by the caller, before the fork."""
    if len(branches) == 1 or not torch.cuda.is_available():
        for b in branches:
            b()
        return
    from .. import streams
    cur = torch.cuda.current_stream()
    pool = [s for s in (streams.get(f"tree{i}", cur.device) for i in range(len(branches) - 1)) if s.cuda_stream != cur.cuda_stream]
    if len(pool) < len(branches) - 1:                      # (the caller already runs on a tree stream: the surplus branches in line)
        for b in branches[len(pool) + 1:]:
            b()
        branches = branches[:len(pool) + 1]
    for b, s in zip(branches[1:], pool):
        s.wait_stream(cur)
        with torch.cuda.stream(s):
            b()
    branches[0]()
    for _, s in zip(branches[1:], pool):
        cur.wait_stream(s)


def _f32c(t):
    t = t if t.dtype == torch.float32 else t.float()
    return t if t.is_contiguous() else t.contiguous()


def _bfs(tree, B, dev):
    """Breadth-first order of a spanning tree: sorted index / parent / children per node and the level boundaries."""
    V = tree.H * tree.W
    with torch.no_grad():
        sidx = torch.empty((B, V), dtype=torch.int32, device=dev)
        spar = torch.empty((B, V), dtype=torch.int32, device=dev)
        schild = torch.empty((B, V, 4), dtype=torch.int32, device=dev)
        levels = torch.empty((B, V + 2), dtype=torch.int32, device=dev)
        L.tree_bfs(tree.edges, tree.H, tree.W, sidx, spar, schild, levels)
    return sidx, spar, schild, levels


class MinimumSpanningTree(nn.Module):
    def __init__(self, distance_func=None):
        super().__init__()
        if distance_func is not None and distance_func is not TreeFilter2D.norm2_distance:
            raise NotImplementedError("only the squared-L2 feature distance (TreeFilter2D.norm2_distance) is built")
        self.distance_func = distance_func

    def forward(self, guide_in, label=None):
        if label is not None:
            raise NotImplementedError("MinimumSpanningTree(label=...) is not used by the tree-energy losses")
        with torch.no_grad():
            g = _f32c(guide_in.detach())
            B, C, H, W = g.shape
            V, E = H * W, 2 * H * W - H - W
            weight = torch.empty((B, E), dtype=torch.float32, device=g.device)
            L.tree_grid_weights(g, weight)
            edges = torch.empty((B, V - 1, 2), dtype=torch.int32, device=g.device)
            L.tree_mst(weight, H, W, edges)
        return Tree(edges, H, W)

    def forward_many(self, guides):
        """The spanning trees AND breadth-first orders of several guidance maps, built side by side (they are independent:
        the multi-scale loss needs four of them before its filter chain starts).  -> [Tree] with `.bfs` attached."""
        trees = [None] * len(guides)

        def build(i):
            def run():
                t = self.forward(guides[i])
                t.bfs = _bfs(t, guides[i].shape[0], guides[i].device)
                trees[i] = t
            return run

        _parallel(*[build(i) for i in range(len(guides))])
        return trees


class _Refine(Function):
    """refine_forward / refine_backward_feature / refine_backward_weight (src/refine/refine.cu:201-370)."""

    @staticmethod
    def forward(ctx, feature_in, edge_weight, sidx, spar, schild, levels, low_tree):
        B, C, V = feature_in.shape
        dev = feature_in.device
        up = torch.empty_like(feature_in)
        aggr = torch.empty_like(feature_in)
        wup = torch.empty((B, 1, V), dtype=torch.float32, device=dev)
        wsum = torch.empty((B, 1, V), dtype=torch.float32, device=dev)

        def features():
            L.tree_aggr_up(feature_in, edge_weight, sidx, schild, levels, up)
            L.tree_prop_down(up, edge_weight, sidx, spar, levels, aggr)

        def normaliser():
            L.tree_aggr_up(None, edge_weight, sidx, schild, levels, wup)
            L.tree_prop_down(wup, edge_weight, sidx, spar, levels, wsum)

        _parallel(features, normaliser)
        out = aggr / wsum
        ctx.save_for_backward(edge_weight, sidx, spar, schild, levels, out, aggr, up, wsum, wup)
        ctx.low_tree = low_tree
        return out

    @staticmethod
    def backward(ctx, grad_output):
        edge_weight, sidx, spar, schild, levels, out, aggr, up, wsum, wup = ctx.saved_tensors
        g = _f32c(grad_output)
        B, C, V = g.shape
        gn = (g / wsum).contiguous()
        gn_up = torch.empty_like(gn)
        grad_feature = torch.empty_like(gn)
        if ctx.low_tree:
            L.tree_aggr_up(gn, edge_weight, sidx, schild, levels, gn_up)
            L.tree_prop_down(gn_up, edge_weight, sidx, spar, levels, grad_feature)
            return grad_feature, None, None, None, None, None, None
        fg = (gn * out).contiguous()
        fg_up = torch.empty_like(fg)
        g_all = torch.empty_like(gn)
        g_norm = torch.empty_like(gn)

        def through_features():
            L.tree_aggr_up(gn, edge_weight, sidx, schild, levels, gn_up)
            L.tree_prop_down(gn_up, edge_weight, sidx, spar, levels, grad_feature)
            L.tree_grad_rec(up, gn_up, aggr, edge_weight, sidx, spar, levels, g_all)       # gn_up is propagated in place

        def through_normaliser():
            L.tree_aggr_up(fg, edge_weight, sidx, schild, levels, fg_up)
            L.tree_grad_rec(wup, fg_up, wsum, edge_weight, sidx, spar, levels, g_norm)

        _parallel(through_features, through_normaliser)
        grad_weight = (g_all - g_norm).sum(1)
        return grad_feature, grad_weight, None, None, None, None, None


class _EdgeWeight(Function):
    """exp(-|e_i - e_parent|^2 * inv_sigma) per tree edge, in BFS order (TreeFilter2D.build_edge_weight)."""

    @staticmethod
    def forward(ctx, embed, sidx, spar, schild, inv_sigma):
        B, Ce, V = embed.shape
        w = torch.empty((B, V), dtype=torch.float32, device=embed.device)
        L.tree_edge_weights(embed, sidx, spar, inv_sigma, w)
        ctx.save_for_backward(embed, sidx, spar, schild, w)
        ctx.inv_sigma = inv_sigma
        return w

    @staticmethod
    def backward(ctx, gw):
        embed, sidx, spar, schild, w = ctx.saved_tensors
        ge = torch.empty_like(embed)
        L.tree_edge_weights_bwd(embed, sidx, spar, schild, w, _f32c(gw), ctx.inv_sigma, ge)
        return ge, None, None, None, None


class TreeFilter2D(nn.Module):
    def __init__(self, groups=1, sigma=0.02, distance_func=None, enable_log=False):
        super().__init__()
        if groups != 1 or distance_func is not None:
            raise NotImplementedError("TreeFilter2D: groups = 1 and the default squared-L2 distance (all the losses use)")
        self.groups, self.sigma, self.enable_log = groups, sigma, enable_log
        self.distance_func = self.norm2_distance

    @staticmethod
    def norm2_distance(fm_ref, fm_tar):
        diff = fm_ref - fm_tar
        return (diff * diff).sum(dim=1)

    def forward(self, feature_in, embed_in, tree, low_tree=True):
        ori_shape = feature_in.shape
        B, C, H, W = ori_shape
        V = H * W
        dev = feature_in.device
        sidx, spar, schild, levels = getattr(tree, "bfs", None) or _bfs(tree, B, dev)
        emb = _f32c(embed_in).reshape(B, embed_in.shape[1], V)
        inv_sigma = 1.0 / self.sigma if low_tree else 1.0
        if low_tree:
            with torch.no_grad():                                   # refine returns no weight gradient for the low tree
                w = _EdgeWeight.apply(emb.detach(), sidx, spar, schild, inv_sigma)
        else:
            w = _EdgeWeight.apply(emb, sidx, spar, schild, inv_sigma)
        out = _Refine.apply(_f32c(feature_in).reshape(B, C, V), w, sidx, spar, schild, levels, low_tree)
        return out.reshape(ori_shape)
