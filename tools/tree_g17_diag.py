#!/usr/bin/env python
"""golden g17's multi-scale tree-energy loss on the HIP path with the HIP glue (csrc/treeglue.hip) and with the torch glue of
rounds 1-4: how far are the head-map gradients from the reference's vectors, per map (GPU box)?  Beside it: how many elements of
each resized guidance map differ from torch's CPU interpolation (the reference's), and by how much."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import torch.nn.functional as F


def main():
    from fedicra_amd import tree_energy as TE
    g = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "g17_tree_glue.npz"))
    dev = "cuda"
    low = torch.from_numpy(g["image"]).repeat(1, 3, 1, 1).to(dev)
    unl = torch.from_numpy(g["roi"]).to(dev)
    for glue in (False, True):
        TE._GLUE = glue
        t = {k: torch.from_numpy(g[k].copy()).to(dev).requires_grad_(True) for k in ("preds", "h1", "h2", "h3")}
        loss, a1, a2, a3 = TE.MScaleRecurveTreeEnergyLoss()(t["preds"], low, t["h1"], t["h2"], t["h3"], unl, 0.6)
        loss.backward()
        torch.cuda.synchronize()
        print(f"glue={'hip' if glue else 'torch'}: loss {loss.item():.8f} (reference {float(g['ms/loss']):.8f})")
        for k in ("preds", "h1", "h2", "h3"):
            d = np.abs(t[k].grad.detach().float().cpu().numpy() - g["ms/g_" + k])
            m = float(np.abs(g["ms/g_" + k]).max())
            print(f"    d{k}: mean|d| / max|g| {d.mean() / m:.2e}   max|d| / max|g| {d.max() / m:.2e}")
        for a, k in ((a1, "AS1"), (a2, "AS2"), (a3, "AS3")):
            d = np.abs(a.detach().float().cpu().numpy() - g["ms/" + k])
            print(f"    {k}: mean {d.mean():.2e} max {d.max():.2e}")
        # the resized guidance maps against torch CPU
        S = t["preds"].shape[-1]
        for k in ("h1", "h2"):
            src = torch.from_numpy(g[k].copy())
            ref = F.interpolate(src, size=(S, S), mode="bilinear", align_corners=False)
            if glue:
                out = TE._TreePrep.apply(t["preds"].detach(), low, unl, src.to(dev))[4].cpu()
            else:
                out = F.interpolate(src.to(dev), size=(S, S), mode="bilinear", align_corners=False).cpu()
            print(f"    resized {k}: {(out != ref).sum().item()} of {ref.numel()} elements differ from torch CPU, max {float((out - ref).abs().max()):.2e}")
    TE._GLUE = True


if __name__ == "__main__":
    main()
