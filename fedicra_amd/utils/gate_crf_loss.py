"""Gated CRF loss with the reference's class surface (/root/reference/code/utils/gate_crf_loss.py), on one fused HIP
kernel (fi_gatedcrf_fwd): nothing is unfolded -- the reference materialises N*C*(2r+1)^2*H*W floats per call.

Supported: what the reference's trainer uses (flower_pCE_2D_GateCRFMsacleTreeEnergyLoss_Ours.py:68-69,143-150): Potts
compatibility, no source / destination masks, modalities at the prediction's resolution, kernels built from 'xy' and/or
the sample.  Anything else raises NotImplementedError instead of silently taking another path."""
from __future__ import annotations

import torch
from torch.autograd import Function

from .. import _lib as L


class _GatedCRF(Function):
    @staticmethod
    def forward(ctx, y_nhwc, feat_nhwc, radius, weights, sxy, ss):
        N, H, W, C = y_nhwc.shape
        prod = torch.empty_like(y_nhwc)
        acc = torch.zeros(2 * L.CRF_SLOTS, dtype=torch.float64, device=y_nhwc.device)
        L.gatedcrf_fwd(y_nhwc, feat_nhwc, radius, weights, sxy, ss, prod, acc)
        a = acc.view(L.CRF_SLOTS, 2).sum(0)
        ctx.save_for_backward(prod)
        ctx.denom = float(N * H * W)
        return ((a[0] - a[1]) / ctx.denom).float()

    @staticmethod
    def backward(ctx, g):
        (prod,) = ctx.saved_tensors
        return prod * (g * (-2.0 / ctx.denom)), None, None, None, None, None


class ModelLossSemsegGatedCRF(torch.nn.Module):
    def forward(self, y_hat_softmax, kernels_desc, kernels_radius, sample, height_input, width_input, mask_src=None,
                mask_dst=None, compatibility=None, custom_modality_downsamplers=None, out_kernels_vis=False):
        assert y_hat_softmax.dim() == 4, "Prediction must be a NCHW batch"
        N, C, height_pred, width_pred = y_hat_softmax.shape
        assert width_input % width_pred == 0 and height_input % height_pred == 0 and \
            width_input * height_pred == height_input * width_pred, \
            f"[{width_input}x{height_input}] !~= [{width_pred}x{height_pred}]"
        if mask_src is not None or mask_dst is not None or compatibility is not None or out_kernels_vis \
                or custom_modality_downsamplers is not None:
            raise NotImplementedError("gated CRF: masks / compatibility matrix / kernel visualisation are not used by the "
                                      "FedICRA trainer and are not built")
        if tuple(sample.shape[2:]) != (height_pred, width_pred):
            raise NotImplementedError("gated CRF: modality at a different resolution than the prediction")
        weights, sxy, ss = [], [], []
        for desc in kernels_desc:
            extra = set(desc) - {"weight", "xy"}
            if len(extra) > 1:
                raise NotImplementedError("gated CRF: one sample modality per kernel (the reference passes one tensor)")
            weights.append(desc["weight"])
            sxy.append(desc.get("xy", 0.0))
            ss.append(desc[next(iter(extra))] if extra else 0.0)
        y = y_hat_softmax.permute(0, 2, 3, 1)
        y = y.float() if y.dtype != torch.float32 else y
        f = sample.permute(0, 2, 3, 1).float().contiguous()
        loss = _GatedCRF.apply(y.contiguous(), f, int(kernels_radius), weights, sxy, ss)
        return {"loss": loss}
