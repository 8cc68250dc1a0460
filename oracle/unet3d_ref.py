"""CPU restatement of the reference's 3D U-Net surface.  TEST INFRASTRUCTURE ONLY:
imported by tests/ and oracle/gen_golden.py, never by fedicra_amd/.

Follows /root/reference/code/networks/unet_3D.py:20-94 (wiring, dropout placement, final 1x1x1 conv) and
/root/reference/code/networks/utils.py:99-123 (UnetConv3 = [Conv3d 3^3 pad 1 -> InstanceNorm3d(affine=False) -> ReLU] x 2),
:260-276 (UnetUp3_CT = trilinear x2 (align_corners False) -> pad skip -> cat([skip, up]) -> UnetConv3), written with
torch.nn.functional calls.  Pinned against the reference's own module by tests/golden/g9_unet3d.npz
(oracle/gen_golden.py:g9_unet3d): same state_dict keys, logits and parameter gradients.
"""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F


class RefUnetConv3(nn.Module):
    def __init__(self, cin, cout):
        super().__init__()
        # same container indices as the reference: {conv1,conv2}.0 = Conv3d, .1 = InstanceNorm3d, .2 = ReLU
        self.conv1 = nn.Sequential(nn.Conv3d(cin, cout, 3, 1, 1), nn.InstanceNorm3d(cout), nn.ReLU(inplace=True))
        self.conv2 = nn.Sequential(nn.Conv3d(cout, cout, 3, 1, 1), nn.InstanceNorm3d(cout), nn.ReLU(inplace=True))

    @staticmethod
    def _half(seq, x):
        y = F.conv3d(x, seq[0].weight, seq[0].bias, stride=1, padding=1)
        return F.relu(F.instance_norm(y, eps=1e-5))

    def forward(self, x):
        return self._half(self.conv2, self._half(self.conv1, x))


class RefUnetUp3CT(nn.Module):
    def __init__(self, cin, cout):
        super().__init__()
        self.conv = RefUnetConv3(cin + cout, cout)

    def forward(self, skip, low):
        up = F.interpolate(low, scale_factor=(2, 2, 2), mode="trilinear", align_corners=False)
        off = up.size(2) - skip.size(2)
        skip = F.pad(skip, 2 * [off // 2, off // 2, 0])
        return self.conv(torch.cat([skip, up], 1))


class RefUNet3D(nn.Module):
    def __init__(self, feature_scale=4, n_classes=21, is_deconv=True, in_channels=3, is_batchnorm=True):
        super().__init__()
        assert is_batchnorm, "the reference always builds the InstanceNorm variant"
        f = [int(c / feature_scale) for c in (64, 128, 256, 512, 1024)]
        self.conv1, self.conv2 = RefUnetConv3(in_channels, f[0]), RefUnetConv3(f[0], f[1])
        self.conv3, self.conv4 = RefUnetConv3(f[1], f[2]), RefUnetConv3(f[2], f[3])
        self.center = RefUnetConv3(f[3], f[4])
        self.up_concat4, self.up_concat3 = RefUnetUp3CT(f[4], f[3]), RefUnetUp3CT(f[3], f[2])
        self.up_concat2, self.up_concat1 = RefUnetUp3CT(f[2], f[1]), RefUnetUp3CT(f[1], f[0])
        self.final = nn.Conv3d(f[0], n_classes, 1)
        self.p_drop = 0.3
        self.dropout_masks = None        # optional [mask_center, mask_up1] (keep masks) for parity runs

    def _drop(self, x, which):
        if not self.training:
            return x
        if self.dropout_masks is not None:
            return x * self.dropout_masks[which].to(x.dtype) / (1.0 - self.p_drop)
        return F.dropout(x, self.p_drop, True)

    def forward(self, x):
        c1 = self.conv1(x)
        c2 = self.conv2(F.max_pool3d(c1, 2))
        c3 = self.conv3(F.max_pool3d(c2, 2))
        c4 = self.conv4(F.max_pool3d(c3, 2))
        ce = self._drop(self.center(F.max_pool3d(c4, 2)), 0)
        u4 = self.up_concat4(c4, ce)
        u3 = self.up_concat3(c3, u4)
        u2 = self.up_concat2(c2, u3)
        u1 = self._drop(self.up_concat1(c1, u2), 1)
        return F.conv3d(u1, self.final.weight, self.final.bias)


class RefUNet3DLC(RefUNet3D):
    """CPU restatement of fedicra_amd.networks.unet_3D.unet_3D_lc (BASELINE configs[4]; the reference has no such model:
    this is the 2D UNet_LC's channel selection (unet.py:103-144) and auxiliary head (unet.py:243-285) written for volumes
    with plain torch calls -- the definition the HIP model is tested against, "parity unpinned" against the reference by
    construction)."""

    def __init__(self, feature_scale=4, n_classes=2, in_channels=1, client_num=8, client_id=0, head_width=4):
        super().__init__(feature_scale, n_classes, True, in_channels, True)
        from .unet_ref import RefPCS
        f = [int(c / feature_scale) for c in (64, 128, 256, 512, 1024)]
        hid = head_width * f[1]
        self.dsn_head = nn.Sequential(nn.Conv3d(f[1], hid, 3, 1, 1), nn.InstanceNorm3d(hid), nn.ReLU(inplace=True),
                                      nn.Dropout3d(0.10), nn.Conv3d(hid, n_classes, 1, bias=False))
        self.pcs_list = [RefPCS(f[4], client_num)]
        self.n_client, self.cid = client_num, client_id
        self.head_mask = None            # optional keep mask [B, hid, 1, 1, 1] for parity runs

    def forward(self, x, emb_idx=None):
        who = self.cid if not emb_idx else emb_idx
        c1 = self.conv1(x)
        c2 = self.conv2(F.max_pool3d(c1, 2))
        c3 = self.conv3(F.max_pool3d(c2, 2))
        c4 = self.conv4(F.max_pool3d(c3, 2))
        ce = self.center(F.max_pool3d(c4, 2))
        emb = torch.zeros(x.size(0), self.n_client, dtype=x.dtype)
        emb[:, who] = 1
        B, C = ce.shape[:2]
        # the 2D module on the volume folded to [B, C, D*H, W]: its pooling is global, the gate per channel
        y2, h = self.pcs_list[0](ce.reshape(B, C, -1, ce.shape[-1]), emb)
        ce = self._drop(y2.reshape(ce.shape), 0)
        u4 = self.up_concat4(c4, ce)
        u3 = self.up_concat3(c3, u4)
        u2 = self.up_concat2(c2, u3)
        u1 = self._drop(self.up_concat1(c1, u2), 1)
        out = F.conv3d(u1, self.final.weight, self.final.bias)
        z = F.relu(F.instance_norm(F.conv3d(u2, self.dsn_head[0].weight, self.dsn_head[0].bias, padding=1), eps=1e-5))
        if self.training:
            z = z * self.head_mask.to(z.dtype) / 0.9 if self.head_mask is not None else F.dropout3d(z, 0.1, True)
        aux = F.conv3d(z, self.dsn_head[4].weight)
        return [out, [c1, c2, c3, c4, ce], u4, u3, u2, u1, [None, None, None, None, h.reshape(B, C, 1, 1, 1)], aux]
