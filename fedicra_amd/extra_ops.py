"""Operators of the reference's module surface that none of its training paths ever runs:
``nn.ConvTranspose2d / ConvTranspose3d(kernel_size=2, stride=2)`` -- UpBlock(bilinear=False) (networks/unet.py:60-62) and
VNet's UpsamplingDeconvBlock (networks/vnet.py:94-118) -- and ``nn.GroupNorm`` (vnet.py:5-31).  Correctness first: the
transposed convolution is one 1x1 implicit-GEMM launch to P*Cout channels plus a depth-to-space shuffle (csrc/extra.hip);
GroupNorm is one workgroup per (sample, group).  Tensors are dense channel-last ([N,H,W,C] / [N,D,H,W,C]) in the compute
dtype, parameters fp32, like everywhere else in fedicra_amd.ops."""
from __future__ import annotations

import torch
from torch.autograd import Function

from . import _lib as L
from . import ops


def _accumulate(param, grad):
    """Add `grad` (logical parameter shape) into the parameter's gradient: the flat-store sink if it has one."""
    tgt, _fresh, ret = ops._grad_target(param)
    tgt.add_(grad.to(tgt.dtype))
    return ret


class _ConvTranspose2x(Function):
    @staticmethod
    def forward(ctx, x, weight, bias):
        three_d = x.dim() == 5
        if three_d:
            N, D, H, W, cin = x.shape
        else:
            (N, H, W, cin), D = x.shape, 1
        P = 8 if three_d else 4
        cout = weight.shape[1]
        dev, dt = x.device, x.dtype
        # rows ordered [tap][co]: W'[(p, co)][ci] = weight[ci][co][tap p]
        w2 = (weight.detach().permute(2, 3, 4, 1, 0) if three_d else weight.detach().permute(2, 3, 1, 0)).reshape(P * cout, cin)
        w2 = w2.contiguous().float()
        wp = torch.empty(P * cout * cin, dtype=dt, device=dev)
        L.pack_weights(w2.view(P * cout, 1, 1, cin), wp, P * cout, 1, cin, 0)
        b2 = None if bias is None else bias.detach().float().repeat(P).contiguous()
        x2 = x.reshape(N * D, H, W, cin)
        tmp = torch.empty((N * D, H, W, P * cout), dtype=dt, device=dev)
        y = torch.empty((N, 2 * D, 2 * H, 2 * W, cout) if three_d else (N, 2 * H, 2 * W, cout), dtype=dt, device=dev)
        L.convtranspose2x_fwd(x2, wp, b2, tmp, y, N, D, H, W, cin, cout, three_d)
        ctx.save_for_backward(x2, w2)
        ctx.meta = (N, D, H, W, cin, cout, P, three_d, weight, bias, tuple(x.shape))
        return y

    @staticmethod
    def backward(ctx, dy):
        x2, w2 = ctx.saved_tensors
        N, D, H, W, cin, cout, P, three_d, weight, bias, xshape = ctx.meta
        dev, dt = x2.device, x2.dtype
        dy = dy.contiguous()
        if dy.dtype != dt:
            dy = dy.to(dt)
        dtmp = torch.empty((N * D, H, W, P * cout), dtype=dt, device=dev)
        dx = gw = gb = None
        wt = dx2 = None
        if ctx.needs_input_grad[0]:
            wt = torch.empty(P * cout * cin, dtype=dt, device=dev)
            L.pack_weights(w2.view(P * cout, 1, 1, cin), wt, P * cout, 1, cin, 1)       # dgrad operand of the 1x1 conv
            dx2 = torch.empty((N * D, H, W, cin), dtype=dt, device=dev)
        L.convtranspose2x_dgrad(dy, wt, dtmp, dx2, N, D, H, W, cin, cout, three_d)     # un-shuffle (+ the dgrad GEMM)
        if dx2 is not None:
            dx = dx2.reshape(xshape)
        if ctx.needs_input_grad[1] or (bias is not None and ctx.needs_input_grad[2]):
            dw = torch.zeros((P * cout, 1, 1, cin), dtype=torch.float32, device=dev)
            db = torch.zeros(P * cout, dtype=torch.float32, device=dev)
            L.convtranspose2x_wgrad(x2, dtmp, dw, db, N, D, H, W, cin, cout, three_d)
            if ctx.needs_input_grad[1]:
                g = dw.view((2, 2, 2, cout, cin) if three_d else (2, 2, cout, cin))
                g = g.permute(4, 3, 0, 1, 2) if three_d else g.permute(3, 2, 0, 1)
                gw = _accumulate(weight, g)
            if bias is not None and ctx.needs_input_grad[2]:
                gb = _accumulate(bias, db.view(P, cout).sum(0))
        return dx, gw, gb


def conv_transpose2x(x, mod):
    """nn.ConvTranspose2d / nn.ConvTranspose3d with kernel_size = stride = 2 (anything else raises)."""
    k, s = tuple(mod.kernel_size), tuple(mod.stride)
    if set(k) != {2} or set(s) != {2} or any(mod.padding) or any(mod.output_padding) or mod.groups != 1:
        raise NotImplementedError("only ConvTranspose(kernel_size=2, stride=2), the form the reference's modules declare")
    return _ConvTranspose2x.apply(x, mod.weight, mod.bias)


class _GroupNorm(Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, groups, eps, relu):
        N, Cc = x.shape[0], x.shape[-1]
        pixels = x.numel() // (N * Cc)
        dev = x.device
        g32, b32 = gamma.detach().float().contiguous(), beta.detach().float().contiguous()
        z = torch.empty_like(x)
        mean = torch.empty(N * groups, dtype=torch.float32, device=dev)
        invstd = torch.empty_like(mean)
        L.groupnorm_fwd(x, z, g32, b32, mean, invstd, N, pixels, Cc, groups, eps, relu)
        ctx.save_for_backward(x, z, g32, mean, invstd)
        ctx.meta = (N, pixels, Cc, groups, relu, gamma, beta)
        return z

    @staticmethod
    def backward(ctx, dz):
        x, z, g32, mean, invstd = ctx.saved_tensors
        N, pixels, Cc, groups, relu, gamma, beta = ctx.meta
        dz = dz.contiguous()
        if dz.dtype != x.dtype:
            dz = dz.to(x.dtype)
        dx = torch.empty_like(x)
        dg = torch.zeros(Cc, dtype=torch.float32, device=x.device) if ctx.needs_input_grad[1] else None
        db = torch.zeros(Cc, dtype=torch.float32, device=x.device) if ctx.needs_input_grad[2] else None
        L.groupnorm_bwd(dz, x, z, g32, mean, invstd, dx, dg, db, N, pixels, Cc, groups, relu)
        gg = _accumulate(gamma, dg) if dg is not None else None
        gb = _accumulate(beta, db) if db is not None else None
        return dx, gg, gb, None, None, None        # (dg / db are None for tensors that do not require grad)


def group_norm(x, mod, relu=False):
    """nn.GroupNorm(num_groups, C) on a dense channel-last tensor, optionally fused with the ReLU that follows it."""
    return _GroupNorm.apply(x, mod.weight, mod.bias, int(mod.num_groups), float(mod.eps), bool(relu))


class _AffineAct(Function):
    """BatchNorm (train or eval) or no normalisation at all, followed by ReLU, on a dense channel-last tensor of any rank:
    the 2D path's BN kernels over the flattened pixels, with the batch statistics taken by fi_channel_stats."""

    @staticmethod
    def forward(ctx, y, gamma, beta, bn):
        Cc = y.shape[-1]
        dev = y.device
        pix = y.numel() // Cc
        y4 = y.reshape(1, 1, pix, Cc)
        z = torch.empty_like(y4)
        coef = torch.empty(4, Cc, dtype=torch.float32, device=dev)
        if bn is None:                                      # identity affine: ReLU only
            coef[0].fill_(1.0)
            coef[1].zero_()
            coef[2].zero_()
            coef[3].fill_(1.0)
            L.bn_act_fwd(y4, coef[0], coef[1], z, 0.0, None)
            training = False
        else:
            training = bn.training
            stats = None
            if training:
                stats = torch.zeros(L.STATS_SLOTS * Cc * 2, dtype=torch.float64, device=dev)
                L.channel_stats(y4, stats, pix, Cc)
            L.bn_fused_fwd(y4, z, stats, gamma, beta, bn.running_mean, bn.running_var, bn.num_batches_tracked, bn.momentum,
                           bn.eps, training, coef, 0.0, None)
        ctx.save_for_backward(y4, coef)
        ctx.meta = (training, bn, gamma, beta, tuple(y.shape))
        return z.reshape(y.shape)

    @staticmethod
    def backward(ctx, dz):
        y4, coef = ctx.saved_tensors
        training, bn, gamma, beta, shape = ctx.meta
        Cc = y4.shape[-1]
        dz4 = dz.contiguous().reshape(y4.shape)
        if dz4.dtype != y4.dtype:
            dz4 = dz4.to(y4.dtype)
        sums = torch.zeros(L.STATS_SLOTS * Cc * 2, dtype=torch.float64, device=y4.device)
        if training:
            L.bn_act_bwd_reduce(dz4, y4, coef[0], coef[1], coef[2], coef[3], sums, 0.0, None)
        dy = torch.empty_like(y4)
        dg = db = None
        if bn is not None and training:
            dg = torch.zeros(Cc, dtype=torch.float32, device=y4.device)
            db = torch.zeros(Cc, dtype=torch.float32, device=y4.device)
        L.bn_act_bwd_apply(dz4, y4, coef[0], coef[1], coef[2], coef[3], sums, training, dy, dg, db, 0.0, None)
        gg = _accumulate(gamma, dg) if dg is not None and ctx.needs_input_grad[1] else None
        gb = _accumulate(beta, db) if db is not None and ctx.needs_input_grad[2] else None
        return dy.reshape(shape), gg, gb, None


def norm_relu(y, norm):
    """The `normalization` + ReLU tail of VNet's blocks (networks/vnet.py:15-24): `norm` is None ('none'), a BatchNorm3d,
    a GroupNorm or an InstanceNorm3d module."""
    import torch.nn as nn
    if norm is None:
        return _AffineAct.apply(y, None, None, None)
    if isinstance(norm, nn.BatchNorm3d):
        return _AffineAct.apply(y, norm.weight, norm.bias, norm)
    if isinstance(norm, nn.GroupNorm):
        return group_norm(y, norm, relu=True)
    if isinstance(norm, nn.InstanceNorm3d):                 # affine=False, no running statistics (torch's defaults)
        Cc = y.shape[-1]
        one = torch.ones(Cc, dtype=torch.float32, device=y.device)
        return _GroupNorm.apply(y, one, torch.zeros_like(one), Cc, float(norm.eps), True)
    raise NotImplementedError(type(norm).__name__)
