"""CPU restatement of the reference's 2D U-Net family (TEST INFRASTRUCTURE ONLY).

Follows /root/reference/code/networks/unet.py; every class cites the lines it
restates.  Built from plain torch.nn CPU ops so that it can run on the GPU
box's host cores as the checker and as bench.py's ``cpu_baseline`` ("port").
state_dict() key names and order equal the reference's:
they are the federated wire format.

Reference quirks reproduced on purpose:
  * every decoder builds its UpBlocks with the default ``bilinear=True``
    (unet.py:216-223) -> Conv1x1 + bilinear x2 align_corners=True; the
    ConvTranspose2d branch (unet.py:60-62) exists but is never taken;
  * ``LCEncoder.pcs_list`` is a plain python list (unet.py:172-177): the PCS
    modules are NOT registered -> not in parameters()/state_dict();
  * ``if not emb_idx`` (unet.py:186) makes emb_idx=0 mean "own client id".
"""
from __future__ import annotations

import torch
import torch.nn as nn

FEATURE_CHNS = (16, 32, 64, 128, 256)          # unet.py:554
DROPOUT_P = (0.05, 0.1, 0.2, 0.3, 0.5)         # unet.py:555


def _double_conv(cin: int, cout: int, p: float) -> nn.Sequential:
    # unet.py:19-27 -- indices 0,1,4,5 carry state; LeakyReLU default slope 0.01
    return nn.Sequential(
        nn.Conv2d(cin, cout, 3, padding=1), nn.BatchNorm2d(cout), nn.LeakyReLU(), nn.Dropout(p),
        nn.Conv2d(cout, cout, 3, padding=1), nn.BatchNorm2d(cout), nn.LeakyReLU())


class RefConvBlock(nn.Module):
    """unet.py:14-30."""

    def __init__(self, cin, cout, p):
        super().__init__()
        self.conv_conv = _double_conv(cin, cout, p)

    def forward(self, x):
        return self.conv_conv(x)


class RefDownBlock(nn.Module):
    """unet.py:34-46: MaxPool2d(2) then ConvBlock."""

    def __init__(self, cin, cout, p):
        super().__init__()
        self.maxpool_conv = nn.Sequential(nn.MaxPool2d(2), RefConvBlock(cin, cout, p))

    def forward(self, x):
        return self.maxpool_conv(x)


class RefUpBlock(nn.Module):
    """unet.py:49-70."""

    def __init__(self, c1, c2, cout, p, bilinear=True):
        super().__init__()
        self.bilinear = bilinear
        if bilinear:
            self.conv1x1 = nn.Conv2d(c1, c2, 1)
            self.up = nn.Upsample(scale_factor=2, mode="bilinear", align_corners=True)
        else:
            self.up = nn.ConvTranspose2d(c1, c2, 2, stride=2)
        self.conv = RefConvBlock(2 * c2, cout, p)

    def forward(self, deep, skip):
        if self.bilinear:
            deep = self.conv1x1(deep)
        deep = self.up(deep)
        return self.conv(torch.cat([skip, deep], dim=1))


class RefEncoder(nn.Module):
    """unet.py:73-100."""

    def __init__(self, in_chns):
        super().__init__()
        c, p = FEATURE_CHNS, DROPOUT_P
        self.in_conv = RefConvBlock(in_chns, c[0], p[0])
        self.down1 = RefDownBlock(c[0], c[1], p[1])
        self.down2 = RefDownBlock(c[1], c[2], p[2])
        self.down3 = RefDownBlock(c[2], c[3], p[3])
        self.down4 = RefDownBlock(c[3], c[4], p[4])

    def stages(self):
        return [self.in_conv, self.down1, self.down2, self.down3, self.down4]

    def forward(self, x):
        feats = []
        for st in self.stages():
            x = st(x)
            feats.append(x)
        return feats


class RefPCS(nn.Module):
    """PersonalizedChannelSelection, unet.py:103-144."""

    def __init__(self, f_dim, emb_dim):
        super().__init__()
        self.fc1 = nn.Sequential(nn.Conv2d(emb_dim, f_dim, 1, bias=False), nn.ReLU(),
                                 nn.Conv2d(f_dim, f_dim, 1, bias=False))
        self.fc2 = nn.Sequential(nn.Conv2d(2 * f_dim, f_dim // 16, 1, bias=False), nn.ReLU(),
                                 nn.Conv2d(f_dim // 16, f_dim, 1, bias=False))

    def forward(self, x, emb):
        avg = x.mean(dim=(2, 3), keepdim=True)              # AdaptiveAvgPool2d(1)
        mx = x.amax(dim=(2, 3), keepdim=True)               # AdaptiveMaxPool2d(1)
        e = self.fc1(emb[:, :, None, None])
        h = torch.sigmoid(self.fc2(torch.cat([avg, e], 1)) + self.fc2(torch.cat([mx, e], 1)))
        return x * h + x, h


class RefLCEncoder(nn.Module):
    """unet.py:146-203.  pcs_list deliberately unregistered (quirk 1)."""

    def __init__(self, in_chns, pcs_num, emb_num, client_num, client_id):
        super().__init__()
        c, p = FEATURE_CHNS, DROPOUT_P
        self.in_conv = RefConvBlock(in_chns, c[0], p[0])
        self.down1 = RefDownBlock(c[0], c[1], p[1])
        self.down2 = RefDownBlock(c[1], c[2], p[2])
        self.down3 = RefDownBlock(c[2], c[3], p[3])
        self.down4 = RefDownBlock(c[3], c[4], p[4])
        self.n_pcs, self.n_client, self.cid = pcs_num, client_num, client_id
        self.pcs_list = [RefPCS(c[5 - pcs_num + i], emb_num) for i in range(pcs_num)]

    def forward(self, x, emb_idx=None):
        who = self.cid if not emb_idx else emb_idx          # unet.py:186 (quirk 2)
        emb = torch.zeros(x.size(0), self.n_client, dtype=x.dtype)      # x's dtype: fp64 yardstick runs of the oracle
        emb[:, who] = 1
        stages = [self.in_conv, self.down1, self.down2, self.down3, self.down4]
        feats, hmaps = [], []
        for i, st in enumerate(stages):
            x = st(x)
            h = None
            if i >= len(stages) - self.n_pcs:
                x, h = self.pcs_list[i - len(stages) + self.n_pcs](x, emb)
            feats.append(x)
            hmaps.append(h)
        return feats, hmaps


def _dsn_head(cin, n_class):
    # unet.py:261-267 / 309-329
    return nn.Sequential(nn.Conv2d(cin, 512, 3, stride=1, padding=1), nn.BatchNorm2d(512), nn.ReLU(),
                         nn.Dropout2d(0.10), nn.Conv2d(512, n_class, 1, bias=False))


class RefDecoder(nn.Module):
    """unet.py:206-240 (heads=0), 243-285 (heads=1), 288-346 (heads=3)."""

    def __init__(self, n_class, heads=0):
        super().__init__()
        c = FEATURE_CHNS
        self.up1 = RefUpBlock(c[4], c[3], c[3], 0.0)
        self.up2 = RefUpBlock(c[3], c[2], c[2], 0.0)
        self.up3 = RefUpBlock(c[2], c[1], c[1], 0.0)
        self.up4 = RefUpBlock(c[1], c[0], c[0], 0.0)
        self.out_conv = nn.Conv2d(c[0], n_class, 3, padding=1)
        self.heads = heads
        if heads == 1:
            self.dsn_head = _dsn_head(c[2], n_class)
        elif heads == 3:
            self.dsn_head1 = _dsn_head(c[2], n_class)
            self.dsn_head2 = _dsn_head(c[1], n_class)
            self.dsn_head3 = _dsn_head(c[0], n_class)

    def forward(self, f):
        d1 = self.up1(f[4], f[3])
        d2 = self.up2(d1, f[2])
        d3 = self.up3(d2, f[1])
        d4 = self.up4(d3, f[0])
        out = [self.out_conv(d4), d1, d2, d3, d4]
        if self.heads == 1:
            out.append(self.dsn_head(d2))
        elif self.heads == 3:
            out += [self.dsn_head1(d2), self.dsn_head2(d3), self.dsn_head3(d4)]
        return out


class RefUNet(nn.Module):
    """unet.py:549-566 (heads=0), 640-656 UNet_Head, 659-675 UNet_MultiHead."""

    def __init__(self, in_chns, class_num, heads=0):
        super().__init__()
        self.encoder = RefEncoder(in_chns)
        self.decoder = RefDecoder(class_num, heads)

    def forward(self, x):
        f = self.encoder(x)
        o = self.decoder(f)
        return [o[0], f] + o[1:]


class RefUNetLC(nn.Module):
    """unet.py:678-699 UNet_LC (heads=1), 701-722 UNet_LC_MultiHead (heads=3)."""

    def __init__(self, in_chns, class_num, pcs_num, emb_num, client_num, client_id, heads=1):
        super().__init__()
        self.encoder = RefLCEncoder(in_chns, pcs_num, emb_num, client_num, client_id)
        self.decoder = RefDecoder(class_num, heads)

    def forward(self, x, emb_idx=None):
        f, h = self.encoder(x, emb_idx)
        o = self.decoder(f)
        return [o[0], f] + o[1:5] + [h] + o[5:]


def ref_net_factory(args, net_type="unet", in_chns=1, class_num=3):
    """net_factory.py:6-32 for the model strings on the hot path."""
    if net_type == "unet":
        return RefUNet(in_chns, class_num)
    if net_type == "unet_head":
        return RefUNet(in_chns, class_num, heads=1)
    if net_type == "unet_multihead":
        return RefUNet(in_chns, class_num, heads=3)
    if net_type == "unet_lc":
        return RefUNetLC(in_chns, class_num, 1, args.min_num_clients, args.min_num_clients, args.cid, heads=1)
    if net_type == "unet_lc_multihead":
        return RefUNetLC(in_chns, class_num, 1, args.min_num_clients, args.min_num_clients, args.cid, heads=3)
    return None


# ---------------------------------------------------------------------------
# deterministic weights shared by the reference importer, the oracle and the
# HIP modules: fixtures then only need inputs/outputs, never 7 MB of weights.
# ---------------------------------------------------------------------------
def seeded_state(module: nn.Module, seed: int = 2022, extra: dict | None = None) -> None:
    """Overwrite every state_dict entry (and ``extra`` name->tensor, used for the
    unregistered PCS weights) with values drawn from a numpy Generator keyed by
    (seed, entry name).  Conv weights ~ U(-b, b), b = 1/sqrt(fan_in) (kaiming-
    uniform-like scale so activations stay O(1)); BN weight ~ U(0.5, 1.5), BN
    bias ~ U(-0.1, 0.1), running_mean ~ U(-0.1, 0.1), running_var ~ U(0.5, 1.5),
    num_batches_tracked = 0.
    """
    import zlib
    import numpy as np

    items = dict(module.state_dict())
    if extra:
        items.update(extra)
    with torch.no_grad():
        for name, t in items.items():
            rng = np.random.default_rng([seed, zlib.crc32(name.encode())])
            if t.dtype == torch.int64:
                t.zero_()
                continue
            if name.endswith("running_var"):
                v = rng.uniform(0.5, 1.5, t.shape)
            elif name.endswith("running_mean"):
                v = rng.uniform(-0.1, 0.1, t.shape)
            elif t.dim() in (4, 5):                       # Conv2d / Conv3d weights
                b = 1.0 / np.sqrt(float(np.prod(t.shape[1:])))
                v = rng.uniform(-b, b, t.shape) * 1.7
            elif name.endswith("weight"):
                v = rng.uniform(0.5, 1.5, t.shape)
            else:
                v = rng.uniform(-0.1, 0.1, t.shape)
            t.copy_(torch.from_numpy(np.asarray(v, dtype=np.float32)))


def pcs_named_tensors(model: nn.Module) -> dict:
    """name -> tensor for the unregistered PCS weights of an LC model."""
    out = {}
    enc = model.encoder
    for i, pcs in enumerate(getattr(enc, "pcs_list", [])):
        for k, v in pcs.state_dict().items():
            out[f"encoder.pcs_list.{i}.{k}"] = v
    return out
