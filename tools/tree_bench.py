"""Timing of the regulariser losses of the `_Ours` procedure at the trainer's shape (12 x 2 x 256 x 256)."""
import sys, time, torch
sys.path.insert(0, '.')
from fedicra_amd.tree_energy import MScaleRecurveTreeEnergyLoss
from fedicra_amd.utils.gate_crf_loss import ModelLossSemsegGatedCRF
from fedicra_amd.utils.tree_filter import MinimumSpanningTree, TreeFilter2D
from fedicra_amd import _lib as L
dev = 'cuda'
B, C, S = 12, 2, 256
g = torch.Generator().manual_seed(0)
preds = torch.randn(B, C, S, S, generator=g).to(dev).requires_grad_(True)
img = torch.rand(B, 1, S, S, generator=g).to(dev)
highs = [torch.rand(B, C, S // s, S // s, generator=g).to(dev).requires_grad_(True) for s in (4, 2, 1)]
rois = (torch.rand(B, S, S, generator=g) < 0.95).to(dev)
loss_fn, crf = MScaleRecurveTreeEnergyLoss(), ModelLossSemsegGatedCRF()
def sync_time(fn, n=3):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
def tree_step():
    out = loss_fn(preds, img.repeat(1, 3, 1, 1), highs[0], highs[1], highs[2], rois, 0.4)
    out[0].backward()
def crf_step():
    l = crf(torch.softmax(preds, 1), [{"weight": 1, "xy": 6, "rgb": 0.1}], 5, img, S, S)["loss"]
    l.backward()
print("tree-energy loss fwd+bwd (4 trees): %.2f ms" % sync_time(tree_step))
print("gated CRF loss fwd+bwd:            %.2f ms" % sync_time(crf_step))
mst = MinimumSpanningTree(TreeFilter2D.norm2_distance)
x3 = img.repeat(1, 3, 1, 1)
print("MST alone:                          %.2f ms" % sync_time(lambda: mst(x3)))
tree = mst(x3)
tf = TreeFilter2D()
print("filter fwd (low tree):              %.2f ms" % sync_time(lambda: tf(torch.softmax(preds.detach(), 1), x3, tree)))
V = S * S
sidx = torch.empty((B, V), dtype=torch.int32, device=dev); spar = torch.empty_like(sidx)
schild = torch.empty((B, V, 4), dtype=torch.int32, device=dev); levels = torch.empty((B, V + 2), dtype=torch.int32, device=dev)
print("BFS alone:                          %.2f ms" % sync_time(lambda: L.tree_bfs(tree.edges, S, S, sidx, spar, schild, levels)))
print("tree depth (levels) per image:", levels[:, 0].cpu().tolist())
