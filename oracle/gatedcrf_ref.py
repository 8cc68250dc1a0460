"""CPU restatement of the Gated CRF loss as the reference calls it (Potts model, no masks):
/root/reference/code/utils/gate_crf_loss.py:20-124 (forward), :132-175 (kernel construction), :185-190 (zero-padded
unfold); call site flower_pCE_2D_GateCRFMsacleTreeEnergyLoss_Ours.py:68-69,143-150 (radius 5, xy sigma 6, rgb sigma 0.1).
TEST INFRASTRUCTURE ONLY.  Pinned against the reference's own module by tests/golden/g10_gatedcrf.npz.

    K(n, i, d) = sum_k w_k * exp(-0.5 * |phi_k(i + d) - phi_k(i)|^2),  K(n, i, 0) = 0
    loss       = ( sum K  -  sum_{n,c,i} y(n,c,i) * sum_d K(n,i,d) * y(n,c,i+d) ) / (N*H*W)

where phi_k stacks the (x, y) mesh / sigma_xy and the sample / sigma of kernel k, and a neighbour outside the image has
phi = 0 and y = 0 (F.unfold zero padding: it still adds to sum K)."""
from __future__ import annotations

import torch


def gated_crf_loss(y_hat, kernels_desc, radius, sample):
    N, C, H, W = y_hat.shape
    dev = y_hat.device
    xs = torch.arange(W, dtype=torch.float32, device=dev).view(1, 1, 1, W).expand(N, 1, H, W)
    ys = torch.arange(H, dtype=torch.float32, device=dev).view(1, 1, H, 1).expand(N, 1, H, W)
    pad = lambda t: torch.nn.functional.pad(t, (radius, radius, radius, radius))
    ksum = y_hat.new_zeros(())
    prod = torch.zeros_like(y_hat)
    yp = pad(y_hat)
    feats = []
    for desc in kernels_desc:
        parts = []
        for modality, sigma in desc.items():
            if modality == "weight":
                continue
            parts.append(torch.cat([xs, ys], 1) / sigma if modality == "xy" else sample / sigma)
        feats.append((desc["weight"], torch.cat(parts, 1)))
    for dy in range(-radius, radius + 1):
        for dx in range(-radius, radius + 1):
            if dy == 0 and dx == 0:
                continue
            K = y_hat.new_zeros((N, 1, H, W))
            for w, f in feats:
                fp = pad(f)[:, :, radius + dy:radius + dy + H, radius + dx:radius + dx + W]
                K = K + w * torch.exp((-0.5 * (fp - f) ** 2).sum(1, keepdim=True))
            ksum = ksum + K.sum()
            prod = prod + K * yp[:, :, radius + dy:radius + dy + H, radius + dx:radius + dx + W]
    loss = (ksum - (prod * y_hat).sum()) / (N * H * W)
    return loss, prod
