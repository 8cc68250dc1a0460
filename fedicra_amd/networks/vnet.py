"""VNet with the reference's module surface (/root/reference/code/networks/vnet.py:5-234: ConvBlock, ResidualConvBlock,
DownsamplingConvBlock, UpsamplingDeconvBlock, VNet; `net_factory_3d('vnet')` builds it with batchnorm and dropout).  The
reference never trains it; it is built for the surface, correctness first:

* Conv3d 3x3x3 = the depth-sliced implicit-GEMM launches of ops3d;
* Conv3d(kernel 2, stride 2) = space-to-depth (fi_depth_to_space2x, inverse) + one 1x1 convolution over 8*Cin channels,
  ConvTranspose3d(kernel 2, stride 2) = one 1x1 convolution to 8*Cout channels + depth-to-space (extra_ops);
* BatchNorm3d / GroupNorm(16) / InstanceNorm3d + ReLU = extra_ops.norm_relu; Dropout3d = the per-(sample, channel) mask
  mode of the dropout kernels.

Same constructor arguments, attribute names and state_dict keys as the reference; tensors cross the module boundary as
NCDHW like there and are dense NDHWC in the compute dtype inside."""
from __future__ import annotations

import torch
from torch import nn
from torch.autograd import Function

from .. import _lib as L
from .. import extra_ops, ops, ops3d
from ..flat import FlatStoreMixin
from .unet import _FiModule


def _norm(normalization, ch):
    if normalization == "batchnorm":
        return nn.BatchNorm3d(ch)
    if normalization == "groupnorm":
        return nn.GroupNorm(num_groups=16, num_channels=ch)
    if normalization == "instancenorm":
        return nn.InstanceNorm3d(ch)
    assert normalization == "none"
    return None


class _Stage(_FiModule):
    """Shared plumbing: `self.conv` is the reference's nn.Sequential (so the state_dict keys agree); `_plan` lists
    (conv module, norm module or None, relu?) in execution order."""

    def _build(self, layers):
        ops_, self._plan = [], []
        for conv, norm, relu in layers:
            ops_.append(conv)
            if norm is not None:
                ops_.append(norm)
            if relu:
                ops_.append(nn.ReLU(inplace=True))
            self._plan.append((conv, norm, relu))
        self.conv = nn.Sequential(*ops_)


def _conv_any(x, conv):
    """Conv3d 3^3 pad 1 | Conv3d(k=2, s=2) | ConvTranspose3d(k=2, s=2) | Conv3d 1^3 on NDHWC."""
    if isinstance(conv, nn.ConvTranspose3d):
        return extra_ops.conv_transpose2x(x, conv)
    if tuple(conv.kernel_size) == (2, 2, 2):
        return _StridedConv2.apply(x, conv.weight, conv.bias)
    return ops3d.conv3d(x, None, conv)


class _StridedConv2(Function):
    """Conv3d(kernel 2, stride 2, padding 0): every output voxel sees its own 2x2x2 block -> space-to-depth, then a 1x1
    convolution over the 8*Cin packed channels."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        N, D, H, W, cin = x.shape
        cout = weight.shape[0]
        dev, dt = x.device, x.dtype
        d, h, w = D // 2, H // 2, W // 2
        xp = torch.empty((N * d, h, w, 8 * cin), dtype=dt, device=dev)
        L.depth_to_space2x(x.contiguous(), xp, N, d, h, w, cin, True, inverse=True)
        # packed channel order [c][a][b][ci]  <->  weight[co][ci][c][a][b]
        w2 = weight.detach().permute(0, 2, 3, 4, 1).reshape(cout, 8 * cin).contiguous().float()
        wp = torch.empty(cout * 8 * cin, dtype=dt, device=dev)
        L.pack_weights(w2.view(cout, 1, 1, 8 * cin), wp, cout, 1, 8 * cin, 0)
        y = torch.empty((N * d, h, w, cout), dtype=dt, device=dev)
        L.conv2d_fwd(xp, None, wp, None if bias is None else bias.detach().float(), y, None, None, ksize=1)
        ctx.save_for_backward(xp, w2)
        ctx.meta = (N, d, h, w, cin, cout, weight, bias)
        return y.view(N, d, h, w, cout)

    @staticmethod
    def backward(ctx, dy):
        xp, w2 = ctx.saved_tensors
        N, d, h, w, cin, cout, weight, bias = ctx.meta
        dev, dt = xp.device, xp.dtype
        dy2 = dy.contiguous().reshape(N * d, h, w, cout)
        if dy2.dtype != dt:
            dy2 = dy2.to(dt)
        dx = gw = gb = None
        if ctx.needs_input_grad[0]:
            wt = torch.empty(cout * 8 * cin, dtype=dt, device=dev)
            L.pack_weights(w2.view(cout, 1, 1, 8 * cin), wt, cout, 1, 8 * cin, 1)
            dxp = torch.empty_like(xp)
            L.conv2d_fwd(dy2, None, wt, None, dxp, None, None, ksize=1)
            dx = torch.empty((N, 2 * d, 2 * h, 2 * w, cin), dtype=dt, device=dev)
            L.depth_to_space2x(dxp, dx, N, d, h, w, cin, True)
        if ctx.needs_input_grad[1] or (bias is not None and ctx.needs_input_grad[2]):
            dw = torch.zeros((cout, 1, 1, 8 * cin), dtype=torch.float32, device=dev)
            db = torch.zeros(cout, dtype=torch.float32, device=dev)
            L.conv2d_wgrad(xp, None, dy2, dw, db, ksize=1)
            if ctx.needs_input_grad[1]:
                gw = extra_ops._accumulate(weight, dw.view(cout, 2, 2, 2, cin).permute(0, 4, 1, 2, 3))
            if bias is not None and ctx.needs_input_grad[2]:
                gb = extra_ops._accumulate(bias, db)
        return dx, gw, gb


class ConvBlock(_Stage):
    def __init__(self, n_stages, n_filters_in, n_filters_out, normalization="none"):
        super().__init__()
        self._build([(nn.Conv3d(n_filters_in if i == 0 else n_filters_out, n_filters_out, 3, padding=1),
                      _norm(normalization, n_filters_out), True) for i in range(n_stages)])

    def _run(self, x):
        for conv, norm, _ in self._plan:
            x = extra_ops.norm_relu(_conv_any(x, conv), norm)
        return x

    def forward(self, x):
        return _out5(self._run(_in5(self, x)))


class ResidualConvBlock(_Stage):
    def __init__(self, n_stages, n_filters_in, n_filters_out, normalization="none"):
        super().__init__()
        if normalization != "none":
            raise NotImplementedError("ResidualConvBlock with a normalisation (no ReLU after the last norm) is not built")
        self._build([(nn.Conv3d(n_filters_in if i == 0 else n_filters_out, n_filters_out, 3, padding=1), None,
                      i != n_stages - 1) for i in range(n_stages)])
        self.relu = nn.ReLU(inplace=True)

    def _run(self, x):
        y = x
        for conv, _, relu in self._plan:
            y = _conv_any(y, conv)
            if relu:
                y = extra_ops.norm_relu(y, None)
        return extra_ops.norm_relu(_Add.apply(y, x), None)

    def forward(self, x):
        return _out5(self._run(_in5(self, x)))


class DownsamplingConvBlock(_Stage):
    def __init__(self, n_filters_in, n_filters_out, stride=2, normalization="none"):
        super().__init__()
        assert stride == 2
        self._build([(nn.Conv3d(n_filters_in, n_filters_out, stride, padding=0, stride=stride),
                      _norm(normalization, n_filters_out), True)])

    def _run(self, x):
        conv, norm, _ = self._plan[0]
        return extra_ops.norm_relu(_conv_any(x, conv), norm)

    def forward(self, x):
        return _out5(self._run(_in5(self, x)))


class UpsamplingDeconvBlock(_Stage):
    def __init__(self, n_filters_in, n_filters_out, stride=2, normalization="none"):
        super().__init__()
        assert stride == 2
        self._build([(nn.ConvTranspose3d(n_filters_in, n_filters_out, stride, padding=0, stride=stride),
                      _norm(normalization, n_filters_out), True)])

    def _run(self, x):
        conv, norm, _ = self._plan[0]
        return extra_ops.norm_relu(_conv_any(x, conv), norm)

    def forward(self, x):
        return _out5(self._run(_in5(self, x)))


class _Add(Function):
    """Skip additions (vnet.py:205-220) as an fi_axpy launch."""

    @staticmethod
    def forward(ctx, a, b):
        if a.dtype != torch.float32:                      # fi_axpy is fp32: 16-bit tensors take the cast kernel round trip
            af, bf = a.float(), b.float()
            L.axpy(af, bf, 1.0)
            return af.to(a.dtype)
        out = a.clone()
        L.axpy(out, b.contiguous(), 1.0)
        return out

    @staticmethod
    def backward(ctx, g):
        return g, g


def _in5(mod, x):
    dt = mod.compute_dtype()
    v = x.permute(0, 2, 3, 4, 1)
    return v if (x.dtype == dt and v.is_contiguous()) else v.contiguous().to(dt)


def _out5(y):
    return y.permute(0, 4, 1, 2, 3)


class VNet(FlatStoreMixin, _FiModule):
    def __init__(self, n_channels=3, n_classes=2, n_filters=16, normalization="none", has_dropout=False):
        super().__init__()
        self.has_dropout = has_dropout
        nf, nz = n_filters, normalization
        self.block_one = ConvBlock(1, n_channels, nf, normalization=nz)
        self.block_one_dw = DownsamplingConvBlock(nf, 2 * nf, normalization=nz)
        self.block_two = ConvBlock(2, nf * 2, nf * 2, normalization=nz)
        self.block_two_dw = DownsamplingConvBlock(nf * 2, nf * 4, normalization=nz)
        self.block_three = ConvBlock(3, nf * 4, nf * 4, normalization=nz)
        self.block_three_dw = DownsamplingConvBlock(nf * 4, nf * 8, normalization=nz)
        self.block_four = ConvBlock(3, nf * 8, nf * 8, normalization=nz)
        self.block_four_dw = DownsamplingConvBlock(nf * 8, nf * 16, normalization=nz)
        self.block_five = ConvBlock(3, nf * 16, nf * 16, normalization=nz)
        self.block_five_up = UpsamplingDeconvBlock(nf * 16, nf * 8, normalization=nz)
        self.block_six = ConvBlock(3, nf * 8, nf * 8, normalization=nz)
        self.block_six_up = UpsamplingDeconvBlock(nf * 8, nf * 4, normalization=nz)
        self.block_seven = ConvBlock(3, nf * 4, nf * 4, normalization=nz)
        self.block_seven_up = UpsamplingDeconvBlock(nf * 4, nf * 2, normalization=nz)
        self.block_eight = ConvBlock(2, nf * 2, nf * 2, normalization=nz)
        self.block_eight_up = UpsamplingDeconvBlock(nf * 2, nf, normalization=nz)
        self.block_nine = ConvBlock(1, nf, nf, normalization=nz)
        self.out_conv = nn.Conv3d(nf, n_classes, 1, padding=0)
        self.dropout = nn.Dropout3d(p=0.5, inplace=False)
        self._fi_finish_init()

    def _drop(self, x):
        return ops3d.dropout(x, self.dropout.p, self.training, owner=self.dropout, channel=True)

    def forward(self, input, turnoff_drop=False):
        drop = self.has_dropout and not turnoff_drop
        x = _in5(self, input)
        x1 = self.block_one._run(x)
        x2 = self.block_two._run(self.block_one_dw._run(x1))
        x3 = self.block_three._run(self.block_two_dw._run(x2))
        x4 = self.block_four._run(self.block_three_dw._run(x3))
        x5 = self.block_five._run(self.block_four_dw._run(x4))
        if drop:
            x5 = self._drop(x5)
        x6 = self.block_six._run(_Add.apply(self.block_five_up._run(x5), x4))
        x7 = self.block_seven._run(_Add.apply(self.block_six_up._run(x6), x3))
        x8 = self.block_eight._run(_Add.apply(self.block_seven_up._run(x7), x2))
        x9 = self.block_nine._run(_Add.apply(self.block_eight_up._run(x8), x1))
        if drop:
            x9 = self._drop(x9)
        out = ops3d.conv3d(x9, None, self.out_conv, y_f32=True)
        return _out5(out)
