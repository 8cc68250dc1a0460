"""How much do the head-map gradients of the multi-scale tree-energy loss (golden g17) move when the bilinearly resized guidance
maps are rounded differently by ONE ulp on 40 % of their elements -- i.e. by another correct fp32 evaluation of the same formula?
(CPU oracle only.)  A 4x4 map upsampled x4 is piecewise linear: neighbouring differences tie in exact arithmetic, the last bit picks
the spanning tree.  Measured: mean |d| 2e-3 .. 8e-3 and max |d| 5e-2 .. 9e-2 of max |g| for the 1/4-resolution map."""
import sys
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import numpy as np, torch, torch.nn.functional as F
from oracle import tree_ref as T
g = np.load(__import__("os").path.join(__import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))), "tests", "golden", "g17_tree_glue.npz"))
low = torch.from_numpy(g["image"]).repeat(1, 3, 1, 1)
unl = torch.from_numpy(g["roi"])
def run(perturb):
    t = {k: torch.from_numpy(g[k].copy()).requires_grad_(True) for k in ("preds", "h1", "h2", "h3")}
    orig = F.interpolate
    def interp(x, size=None, mode="nearest", align_corners=None, **k):
        y = orig(x, size=size, mode=mode, align_corners=align_corners, **k)
        if perturb and mode == "bilinear" and x.shape[-1] != size[-1]:
            gen = torch.Generator().manual_seed(perturb)
            ulp = torch.where(torch.rand(y.shape, generator=gen) < 0.4, torch.sign(torch.randn(y.shape, generator=gen)), torch.zeros(y.shape))
            y = y + ulp * (y.abs() * 2.0 ** -24)          # +-1 ulp-ish on 40 % of the elements: another correct rounding of the same formula
        return y
    F.interpolate = interp
    try:
        loss, a1, a2, a3 = T.mscale_recurve_tree_energy_loss(t["preds"], low, t["h1"], t["h2"], t["h3"], unl, 0.6)
        loss.backward()
    finally:
        F.interpolate = orig
    return float(loss), {k: t[k].grad.numpy().copy() for k in t}, [a.detach().numpy() for a in (a1, a2, a3)]
l0, g0, a0 = run(0)
print("golden check: loss", l0, float(g["ms/loss"]), "max |dg_h1 - golden|", float(np.abs(g0["h1"] - g["ms/g_h1"]).max()))
for p in (1, 2, 3):
    l1, g1, a1 = run(p)
    print(f"perturbation {p}: dloss {abs(l1 - l0):.2e}; " + "; ".join(
        f"{k}: mean|d| {np.abs(g1[k] - g0[k]).mean():.2e} = {np.abs(g1[k] - g0[k]).mean() / np.abs(g0[k]).max():.2e} of max|g|, max|d|/max|g| {np.abs(g1[k] - g0[k]).max() / np.abs(g0[k]).max():.2e}"
        for k in ("preds", "h1", "h2", "h3")) + f"; AS3 max|d| {np.abs(a1[2] - a0[2]).max():.2e}")
