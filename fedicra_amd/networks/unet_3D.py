"""3D U-Net surface of the reference, on the HIP kernels.

Mirrors /root/reference/code/networks/unet_3D.py:20-94 and networks/utils.py:99-123 (UnetConv3), :260-276 (UnetUp3_CT):
same class names, constructor arguments, state_dict keys (38 tensors for the default widths: every Conv3d's weight and
bias; InstanceNorm3d carries no state) and the same initialisation calls in the same order (kaiming-normal weights,
networks/networks_other.py:40-49,64-75), so a fixed torch seed yields the reference's initial state.  forward takes the
reference's NCDHW fp32 tensor and returns NCDHW fp32 logits; inside, volumes are dense NDHWC in the compute dtype and
every op is a libfedicra_hip.so launch (fedicra_amd/ops3d.py).  The reference never trains this model (no 3D training
loop exists there); gradients are provided and tested so that the 2D client code can drive it.
"""
from __future__ import annotations

import torch
import torch.nn as nn
from torch.nn import init

from .. import ops, ops3d
from ..flat import FlatStoreMixin
from .unet import _DTYPES, PersonalizedChannelSelection, default_compute_dtype


def _weights_init_kaiming(m):
    """networks_other.py:40-49."""
    name = m.__class__.__name__
    if name.find("Conv") != -1 or name.find("Linear") != -1:
        init.kaiming_normal_(m.weight.data, a=0, mode="fan_in")
    elif name.find("BatchNorm") != -1:
        init.normal_(m.weight.data, 1.0, 0.02)
        init.constant_(m.bias.data, 0.0)


def init_weights(net, init_type="normal"):
    if init_type != "kaiming":
        raise NotImplementedError("only the 'kaiming' initialisation is used on this surface")
    net.apply(_weights_init_kaiming)


class UnetConv3(nn.Module):
    """[Conv3d -> InstanceNorm3d -> ReLU] x 2 (networks/utils.py:99-123)."""

    def __init__(self, in_size, out_size, is_batchnorm, kernel_size=(3, 3, 1), padding_size=(1, 1, 0),
                 init_stride=(1, 1, 1)):
        super().__init__()
        if not is_batchnorm:
            raise NotImplementedError("is_batchnorm=False is never instantiated by the reference")
        if tuple(kernel_size) != (3, 3, 3) or tuple(padding_size) != (1, 1, 1) or tuple(init_stride) != (1, 1, 1):
            raise NotImplementedError("unet_3D only builds 3x3x3 / pad 1 / stride 1 blocks")
        self.conv1 = nn.Sequential(nn.Conv3d(in_size, out_size, kernel_size, init_stride, padding_size),
                                   nn.InstanceNorm3d(out_size), nn.ReLU(inplace=True))
        self.conv2 = nn.Sequential(nn.Conv3d(out_size, out_size, kernel_size, 1, padding_size),
                                   nn.InstanceNorm3d(out_size), nn.ReLU(inplace=True))
        for m in self.children():
            init_weights(m, init_type="kaiming")

    def _run(self, x0, x1=None):
        z = ops3d.conv3d(x0, x1, self.conv1[0], norm=True)
        return ops3d.conv3d(z, None, self.conv2[0], norm=True)


class UnetUp3_CT(nn.Module):
    """trilinear x2 -> cat([skip, up]) -> UnetConv3 (networks/utils.py:260-276); the concat is folded into the conv."""

    def __init__(self, in_size, out_size, is_batchnorm=True):
        super().__init__()
        self.conv = UnetConv3(in_size + out_size, out_size, is_batchnorm, kernel_size=(3, 3, 3), padding_size=(1, 1, 1))
        self.up = nn.Upsample(scale_factor=(2, 2, 2), mode="trilinear")
        for m in self.children():
            if m.__class__.__name__.find("UnetConv3") != -1:
                continue
            init_weights(m, init_type="kaiming")

    def _run(self, skip, low):
        up = ops3d.upsample3d2x(low)
        if up.shape[1:4] != skip.shape[1:4]:
            # utils.py:271-274 pads the skip by 2 * [offset // 2, offset // 2, 0] with offset = up - skip.  MaxPool3d floors,
            # so the only offset besides 0 is -1 (an odd skip), for which that rule crops W by 2, H and D by 1: torch.cat then
            # raises in the reference too ("Sizes of tensors must match except in dimension 1") -- same error class here.
            raise RuntimeError("Sizes of tensors must match except in dimension 1: skip {} vs up-sampled {} (odd volume "
                               "size; the reference's padding rule does not reconcile them)".format(
                                   tuple(skip.shape[1:4]), tuple(up.shape[1:4])))
        return self.conv._run(skip, up)


class unet_3D(FlatStoreMixin, nn.Module):

    def __init__(self, feature_scale=4, n_classes=21, is_deconv=True, in_channels=3, is_batchnorm=True):
        super().__init__()
        self.is_deconv, self.in_channels = is_deconv, in_channels
        self.is_batchnorm, self.feature_scale = is_batchnorm, feature_scale
        filters = [int(x / self.feature_scale) for x in [64, 128, 256, 512, 1024]]
        k3, p1 = (3, 3, 3), (1, 1, 1)
        self.conv1 = UnetConv3(self.in_channels, filters[0], self.is_batchnorm, kernel_size=k3, padding_size=p1)
        self.maxpool1 = nn.MaxPool3d(kernel_size=(2, 2, 2))
        self.conv2 = UnetConv3(filters[0], filters[1], self.is_batchnorm, kernel_size=k3, padding_size=p1)
        self.maxpool2 = nn.MaxPool3d(kernel_size=(2, 2, 2))
        self.conv3 = UnetConv3(filters[1], filters[2], self.is_batchnorm, kernel_size=k3, padding_size=p1)
        self.maxpool3 = nn.MaxPool3d(kernel_size=(2, 2, 2))
        self.conv4 = UnetConv3(filters[2], filters[3], self.is_batchnorm, kernel_size=k3, padding_size=p1)
        self.maxpool4 = nn.MaxPool3d(kernel_size=(2, 2, 2))
        self.center = UnetConv3(filters[3], filters[4], self.is_batchnorm, kernel_size=k3, padding_size=p1)
        self.up_concat4 = UnetUp3_CT(filters[4], filters[3], is_batchnorm)
        self.up_concat3 = UnetUp3_CT(filters[3], filters[2], is_batchnorm)
        self.up_concat2 = UnetUp3_CT(filters[2], filters[1], is_batchnorm)
        self.up_concat1 = UnetUp3_CT(filters[1], filters[0], is_batchnorm)
        self.final = nn.Conv3d(filters[0], n_classes, 1)
        self.dropout1 = nn.Dropout(p=0.3)
        self.dropout2 = nn.Dropout(p=0.3)
        for m in self.modules():
            if isinstance(m, nn.Conv3d):
                init_weights(m, init_type="kaiming")
            elif isinstance(m, nn.BatchNorm3d):
                init_weights(m, init_type="kaiming")
        self._fi_dtype = None
        self._fi_finish_init()                  # one flat fp32 state / gradient buffer: fused AdamW, aggregation, ALA

    def compute_dtype(self):
        return self._fi_dtype if self._fi_dtype is not None else default_compute_dtype()

    def set_compute_dtype(self, dtype):
        self._fi_dtype = _DTYPES[dtype] if isinstance(dtype, str) else dtype
        return self

    def _trunk(self, inputs, gate=None):
        """-> (fp32 logits NDHWC, encoder features, decoder features); ``gate(center) -> center`` hooks the site-specific
        channel selection of unet_3D_lc in after the deepest block."""
        dt = self.compute_dtype()
        x = inputs.permute(0, 2, 3, 4, 1).contiguous().to(dt)            # NCDHW -> dense NDHWC (plumbing)
        # (pool_skip3d: an encoder feature is pooled AND concatenated in the up path -- the sum of its two gradients is made by the
        # pooling's backward pass)
        conv1, pooled = ops3d.pool_skip3d(self.conv1._run(x))
        conv2, pooled = ops3d.pool_skip3d(self.conv2._run(pooled))
        conv3, pooled = ops3d.pool_skip3d(self.conv3._run(pooled))
        conv4, pooled = ops3d.pool_skip3d(self.conv4._run(pooled))
        center = self.center._run(pooled)
        if gate is not None:
            center = gate(center)
        center = ops3d.dropout(center, self.dropout1.p, self.training, owner=self.dropout1)
        up4 = self.up_concat4._run(conv4, center)
        up3 = self.up_concat3._run(conv3, up4)
        up2 = self.up_concat2._run(conv2, up3)
        up1 = self.up_concat1._run(conv1, up2)
        up1 = ops3d.dropout(up1, self.dropout2.p, self.training, owner=self.dropout2)
        final = ops3d.conv3d(up1, None, self.final, norm=False, y_f32=True)
        return final, [conv1, conv2, conv3, conv4, center], [up4, up3, up2, up1]

    def forward(self, inputs):
        final, _, _ = self._trunk(inputs)
        return final.permute(0, 4, 1, 2, 3)                              # NCDHW view of the fp32 logits

    @staticmethod
    def apply_argmax_softmax(pred):
        return torch.softmax(pred, dim=1)


def _ncdhw(t):
    return t.permute(0, 4, 1, 2, 3)


class unet_3D_lc(unet_3D):
    """BASELINE.json configs[4]: "3D U-Net + per-client adapter heads".  The reference has no 3D federated model (nothing
    calls net_factory_3d): this is ``unet_3D`` with the two personalisation devices of the reference's
    2D ``UNet_LC`` lifted to volumes, keeping its conventions so that the FedICRA client code drives it unchanged --

      * a ``PersonalizedChannelSelection`` (networks/unet.py:103-144; its fc layers are 1x1 convolutions on pooled vectors,
        so the 2D module is used as it is) on the deepest block (``pcs_num = 1``): one-hot site embedding, global average /
        maximum over the volume, shared MLP, sigmoid gate ``x * h + x``; held in a plain python list like
        ``LCEncoder.pcs_list`` (unet.py:172-177: never optimised, never communicated) with the ``if not emb_idx`` rule
        (unet.py:186: 0 means "own client");
      * an auxiliary adapter head (``Decoder_Head.dsn_head``, unet.py:243-285, as Conv3d 3^3 -> InstanceNorm3d -> ReLU ->
        Dropout3d(0.1) -> Conv3d 1^3 without bias) on the second decoder level;
      * ``forward(x, emb_idx=None)`` returns UNet_LC's list: ``[logits, [conv1..center], up4, up3, up2, up1, heatmaps, aux]``
        with the heat-map ``[B, C, 1, 1, 1]`` last in its list (flower_pCE_2D.py:134 reads ``[6][-1]``).

    Parameters a federation keeps local under FedICRA's naming rule (names containing ``final`` / ``up_concat``, the 3D
    counterparts of ``out_conv`` / ``up1..4``, flower_common.py:506) are listed in ``LOCAL_KEYS``."""

    LOCAL_KEYS = ("final", "up_concat4", "up_concat3", "up_concat2", "up_concat1")

    def __init__(self, feature_scale=4, n_classes=2, is_deconv=True, in_channels=1, is_batchnorm=True, client_num=8,
                 client_id=0, head_width=4):
        unet_3D.__init__(self, feature_scale, n_classes, is_deconv, in_channels, is_batchnorm)
        self.n_client, self.cid = client_num, client_id
        filters = [int(x / feature_scale) for x in [64, 128, 256, 512, 1024]]
        hid = head_width * filters[1]
        self.dsn_head = nn.Sequential(nn.Conv3d(filters[1], hid, 3, 1, 1), nn.InstanceNorm3d(hid), nn.ReLU(inplace=True),
                                      nn.Dropout3d(0.10), nn.Conv3d(hid, n_classes, 1, bias=False))
        init_weights(self.dsn_head[0], init_type="kaiming")
        init_weights(self.dsn_head[4], init_type="kaiming")
        self.pcs_list = [PersonalizedChannelSelection(filters[4], client_num)]      # plain list on purpose (quirk 1)
        for pcs in self.pcs_list:
            for q in pcs.parameters():
                q.requires_grad_(False)
        self._fi_finish_init()

    def _apply(self, fn, *a, **k):
        for pcs in getattr(self, "pcs_list", []):
            pcs._apply(fn, *a, **k)
        return super()._apply(fn, *a, **k)

    def set_compute_dtype(self, dtype):
        super().set_compute_dtype(dtype)
        for pcs in self.pcs_list:
            pcs._fi_dtype = self._fi_dtype
        return self

    def forward(self, inputs, emb_idx=None):
        who = self.cid if not emb_idx else emb_idx                         # unet.py:186 (0 means "own")
        hm = []

        def gate(center):
            B, d, h, w, C = center.shape
            idx = torch.full((B,), int(who), dtype=torch.int32, device=center.device)        # one-hot row `who` as its index
            y, hmap = self.pcs_list[0]._run(center.reshape(B, d * h, w, C), idx)    # pooled over the whole volume
            hm.append(hmap)
            return y.reshape(B, d, h, w, C)
        final, enc, dec = self._trunk(inputs, gate)
        z = ops3d.conv3d(dec[2], None, self.dsn_head[0], norm=True)
        z = ops3d.dropout(z, self.dsn_head[3].p, self.training, owner=self.dsn_head[3], channel=True)
        aux = ops3d.conv3d(z, None, self.dsn_head[4], norm=False, y_f32=True)
        heat = hm[0].reshape(hm[0].shape[0], 1, 1, 1, -1)
        return [_ncdhw(final), [_ncdhw(t) for t in enc]] + [_ncdhw(t) for t in dec] + \
               [[None, None, None, None, _ncdhw(heat)], _ncdhw(aux)]
