"""MI355X-native 2D U-Net family with the reference's module surface.

Drop-in for /root/reference/code/networks/unet.py: same class names, constructor signatures,
``forward`` return lists and ``state_dict()`` keys/shapes (the federated wire format), but every
op underneath is a hand-written gfx950 HIP kernel reached through the C ABI
(include/fedicra_hip.h) -- no ATen conv/BN/pool kernels on the path.

Layout: tensors carry the reference's logical NCHW shape, with channels_last strides (dense NHWC
memory).  Sub-modules can therefore be chained or called individually exactly like the
reference's, and the permutes are zero-copy views.  nn.Conv2d / nn.BatchNorm2d objects are used
only as parameter containers (same default init, same state_dict keys); their ATen forward is
never called.

Reference quirks kept on purpose: UpBlock's dead ConvTranspose2d branch is
never built because every decoder uses the default ``bilinear=True`` (unet.py:216-223);
LCEncoder.pcs_list is a plain list (unregistered PCS weights, unet.py:172-177);
``if not emb_idx`` (unet.py:186).
"""
from __future__ import annotations

import os

import torch
import torch.nn as nn

from .. import ops
from ..flat import FlatStoreMixin

FEATURE_CHNS = [16, 32, 64, 128, 256]
DROPOUT = [0.05, 0.1, 0.2, 0.3, 0.5]

_DTYPES = {"fp32": torch.float32, "float32": torch.float32, "bf16": torch.bfloat16, "bfloat16": torch.bfloat16,
           "fp16": torch.float16, "float16": torch.float16, "half": torch.float16}


def default_compute_dtype():
    return _DTYPES[os.environ.get("FEDICRA_DTYPE", "fp32")]


class _FiModule(nn.Module):
    """Base: knows the compute dtype and converts NCHW-shaped tensors <-> dense NHWC."""

    _fi_dtype = None   # set on the root by set_compute_dtype(); modules fall back to the default

    def compute_dtype(self):
        return self._fi_dtype if self._fi_dtype is not None else default_compute_dtype()

    def _in(self, x):
        """NCHW-shaped tensor (any strides / fp32 or compute dtype) -> dense NHWC compute-dtype tensor."""
        dt = self.compute_dtype()
        v = x.permute(0, 2, 3, 1)
        if x.dtype == dt and v.is_contiguous():
            return v
        if x.requires_grad:                       # differentiable slow path (plumbing only)
            return v.contiguous().to(dt)
        return ops.to_nhwc(x, dt)

    @staticmethod
    def _out(y):
        return y.permute(0, 3, 1, 2)


def set_compute_dtype(model: nn.Module, dtype):
    own = getattr(type(model), "set_compute_dtype", None)
    if callable(own):                                      # the 3D models (unet_3D, unet_3D_lc, VNet) carry their own
        return model.set_compute_dtype(dtype)
    dt = _DTYPES[dtype] if isinstance(dtype, str) else dtype
    for m in model.modules():
        if isinstance(m, _FiModule):
            m._fi_dtype = dt
    for pcs in getattr(getattr(model, "encoder", None), "pcs_list", []):
        for m in pcs.modules():
            if isinstance(m, _FiModule):
                m._fi_dtype = dt
    return model


class ConvBlock(_FiModule):
    """two convolution layers with batch norm and leaky relu (unet.py:14-30)."""

    def __init__(self, in_channels, out_channels, dropout_p):
        super().__init__()
        # indices 0,1,4,5 hold state exactly like the reference's nn.Sequential
        self.conv_conv = nn.Sequential(
            nn.Conv2d(in_channels, out_channels, kernel_size=3, padding=1), nn.BatchNorm2d(out_channels),
            nn.LeakyReLU(), nn.Dropout(dropout_p),
            nn.Conv2d(out_channels, out_channels, kernel_size=3, padding=1), nn.BatchNorm2d(out_channels),
            nn.LeakyReLU())
        self.dropout_p = dropout_p

    def _run(self, x0, x1=None):
        s = self.conv_conv
        z = ops.conv_bn_act(x0, x1, s[0], s[1], s[2].negative_slope, self.dropout_p, "elem")
        return ops.conv_bn_act(z, None, s[4], s[5], s[6].negative_slope, 0.0)

    def _probe(self, s0, s1, groups, pool=False, first=False, store=True):
        """This block inside the batched no-grad forward (ops.probe_*): sources and result are raw activations.  store =
        False: nobody reads the block's output (the last decoder block when no head sits on it) -- its second convolution
        runs for the BatchNorm statistics only and writes nothing."""
        s = self.conv_conv
        if first:
            r = ops.probe_first_conv_bn(s0, s[0], s[1], s[2].negative_slope, groups)
        else:
            r = ops.probe_conv_bn(s0, s1, s[0], s[1], s[2].negative_slope, groups, pool=pool)
        drop = ops._probe_drop(self.dropout_p, s[1], groups)
        return ops.probe_conv_bn(r, None, s[4], s[5], s[6].negative_slope, groups, in_drop=drop, store=store)

    def forward(self, x):
        return self._out(self._run(self._in(x)))


class DownBlock(_FiModule):
    """Downsampling followed by ConvBlock (unet.py:34-46)."""

    def __init__(self, in_channels, out_channels, dropout_p):
        super().__init__()
        self.maxpool_conv = nn.Sequential(nn.MaxPool2d(2), ConvBlock(in_channels, out_channels, dropout_p))

    def _run(self, x):
        return self.maxpool_conv[1]._run(ops.maxpool2(x))

    def _run_skip(self, x):
        """(x for the skip connection, block output): pooling and skip share one backward pass (ops._PoolSkip)."""
        skip, pooled = ops.pool_skip(x)
        return skip, self.maxpool_conv[1]._run(pooled)

    def forward(self, x):
        return self._out(self._run(self._in(x)))


class UpBlock(_FiModule):
    """Upsampling followed by ConvBlock (unet.py:49-70).  The concat is folded into the conv gather."""

    def __init__(self, in_channels1, in_channels2, out_channels, dropout_p, bilinear=True):
        super().__init__()
        self.bilinear = bilinear
        if bilinear:
            self.conv1x1 = nn.Conv2d(in_channels1, in_channels2, kernel_size=1)
            self.up = nn.Upsample(scale_factor=2, mode="bilinear", align_corners=True)
        else:       # unet.py:60-62 -- no decoder of the reference selects it (they all pass the default), built for the surface
            self.up = nn.ConvTranspose2d(in_channels1, in_channels2, kernel_size=2, stride=2)
        self.conv = ConvBlock(in_channels2 * 2, out_channels, dropout_p)

    def _run(self, x1, x2):
        if self.bilinear:
            up = ops.conv1x1_up(x1, self.conv1x1)
        else:
            from .. import extra_ops
            up = extra_ops.conv_transpose2x(x1, self.up)
        return self.conv._run(x2, up)            # == ConvBlock(cat([x2, up], dim=1))

    def forward(self, x1, x2):
        return self._out(self._run(self._in(x1), self._in(x2)))


class Encoder(_FiModule):
    """unet.py:73-100."""

    def __init__(self, params):
        super().__init__()
        self.params = params
        self.in_chns, self.ft_chns = params["in_chns"], params["feature_chns"]
        self.n_class, self.bilinear, self.dropout = params["class_num"], params["bilinear"], params["dropout"]
        assert len(self.ft_chns) == 5
        c, p = self.ft_chns, self.dropout
        self.in_conv = ConvBlock(self.in_chns, c[0], p[0])
        self.down1 = DownBlock(c[0], c[1], p[1])
        self.down2 = DownBlock(c[1], c[2], p[2])
        self.down3 = DownBlock(c[2], c[3], p[3])
        self.down4 = DownBlock(c[3], c[4], p[4])

    def _run(self, x):
        x0 = self.in_conv._run(x)
        x0, x1 = self.down1._run_skip(x0)
        x1, x2 = self.down2._run_skip(x1)
        x2, x3 = self.down3._run_skip(x2)
        x3, x4 = self.down4._run_skip(x3)
        return [x0, x1, x2, x3, x4]

    def forward(self, x):
        return [self._out(f) for f in self._run(self._in(x))]


class PersonalizedChannelSelection(_FiModule):
    """unet.py:103-144.  fc layers are 1x1 convs on [B,C,1,1] vectors -> the same conv kernel."""

    def __init__(self, f_dim, emb_dim):
        super().__init__()
        self.fc1 = nn.Sequential(nn.Conv2d(emb_dim, f_dim, 1, bias=False), nn.ReLU(),
                                 nn.Conv2d(f_dim, f_dim, 1, bias=False))
        self.fc2 = nn.Sequential(nn.Conv2d(f_dim * 2, f_dim // 16, 1, bias=False), nn.ReLU(),
                                 nn.Conv2d(f_dim // 16, f_dim, 1, bias=False))

    @staticmethod
    def _mlp(seq, v0, v1=None):
        """v*: fp32 [B,1,1,C] NHWC vectors (fc math stays fp32: it is ~100 kFLOP)."""
        h = torch.relu(ops.conv2d(v0, v1, seq[0]))
        return ops.conv2d(h, None, seq[2])

    def forward_emb(self, emb):
        return self._mlp(self.fc1, emb.reshape(emb.shape[0], 1, 1, -1).float())

    def _run(self, x, who):
        """x: dense NHWC; who: int32 [B], the embedding index of every image (the reference feeds the one-hot rows of them,
        unet.py:180-184).  The whole gate -- fc1 of the embedding, fc2 of [avg ; e] and [max ; e], sigmoid -- is one launch
        (fi_pcs_gate_fwd); on the CPU (host logic tests) the same arithmetic through the 1x1-convolution helpers."""
        B, H, W, C = x.shape
        avg, mx = ops.global_avgmax(x)                       # [B,C] fp32 each
        if x.is_cuda:
            hmap = ops.pcs_gate(avg, mx, who, self.fc1, self.fc2).reshape(B, 1, 1, C)
        else:
            emb = torch.zeros((B, self.fc1[0].weight.shape[1]), device=x.device)
            emb[torch.arange(B), who.long()] = 1
            e = self.forward_emb(emb)                            # [B,1,1,C]
            a = self._mlp(self.fc2, avg.reshape(B, 1, 1, C), e)  # cat([avg, emb], 1) folded into the gather
            m = self._mlp(self.fc2, mx.reshape(B, 1, 1, C), e)
            hmap = torch.sigmoid(a + m)                          # [B,1,1,C]
        return ops.channel_gate(x, hmap.reshape(B, C)), hmap

    def forward(self, x, emb):
        y, h = self._run(self._in(x), emb.argmax(dim=1).to(torch.int32))
        return self._out(y), self._out(h)


class LCEncoder(_FiModule):
    """unet.py:146-203."""

    def __init__(self, params):
        super().__init__()
        self.params = params
        self.in_chns, self.ft_chns = params["in_chns"], params["feature_chns"]
        self.n_class, self.bilinear, self.dropout = params["class_num"], params["bilinear"], params["dropout"]
        self.n_pcs, self.n_emb = params["pcs_num"], params["emb_num"]
        self.n_client, self.cid = params["client_num"], params["client_id"]
        assert len(self.ft_chns) == 5
        c, p = self.ft_chns, self.dropout
        self.in_conv = ConvBlock(self.in_chns, c[0], p[0])
        self.down1 = DownBlock(c[0], c[1], p[1])
        self.down2 = DownBlock(c[1], c[2], p[2])
        self.down3 = DownBlock(c[2], c[3], p[3])
        self.down4 = DownBlock(c[3], c[4], p[4])
        self.conv_list = [self.in_conv, self.down1, self.down2, self.down3, self.down4]
        # plain python list on purpose: unregistered like the reference (quirk 1)
        self.pcs_list = [PersonalizedChannelSelection(c[5 - self.n_pcs + i], self.n_emb) for i in range(self.n_pcs)]
        for pcs in self.pcs_list:                # never optimised / communicated in the reference: skip their wgrad
            for q in pcs.parameters():
                q.requires_grad_(False)

    def _apply(self, fn, *a, **k):
        for pcs in self.pcs_list:                # the reference moves them with an explicit .cuda() (unet.py:176)
            pcs._apply(fn, *a, **k)
        return super()._apply(fn, *a, **k)

    def _run(self, x, emb_idx=None):
        who = self.cid if not emb_idx else emb_idx           # unet.py:186 (quirk 2: 0 means "own")
        emb = self._who((who,), x.shape[0], x.device)        # one-hot row `who` for every image, as its index
        feats, hmaps = [], []
        n = len(self.conv_list)
        for i, blk in enumerate(self.conv_list):
            if i > 0 and feats and isinstance(blk, DownBlock) and feats[-1] is x:
                # the feature is pooled here AND read by the decoder's skip connection: both gradients in one backward pass
                # (ops._PoolSkip, like Encoder._run; autograd added them with an elementwise launch per level otherwise)
                feats[-1], x = blk._run_skip(x)
            else:
                x = blk._run(x)
            h = None
            if i >= n - self.n_pcs:
                x, h = self.pcs_list[i - n + self.n_pcs]._run(x, emb)
            feats.append(x)
            hmaps.append(h)
        return feats, hmaps

    def _who(self, ids, B, device):
        """int32 [len(ids) * B]: embedding index of every image of a batch of len(ids) groups of B (cached: the training
        step asks for the same few combinations every iteration, and a captured step must not re-create them)."""
        cache = self.__dict__.setdefault("_who_cache", {})
        key = (ids, B, str(device))
        t = cache.get(key)
        if t is None:
            t = cache[key] = torch.tensor([i for i in ids for _ in range(B)], dtype=torch.int32, device=device)
        return t

    def _probe(self, x, emb_ids):
        """The encoder for len(emb_ids) forwards of the SAME batch x under different embeddings, as one batch of groups
        (group g = forward g): every ConvBlock half stays a raw activation, the pooling happens in the consumer's loader,
        only the deepest level is materialised for the channel selection.  -> (4 raw skips, x4 tensor, heat-map [G*B,1,1,C])"""
        G, B = len(emb_ids), x.shape[0]
        r = self.in_conv._probe(x, None, G, first=True)
        feats = [r]
        for blk in (self.down1, self.down2, self.down3, self.down4):
            r = blk.maxpool_conv[1]._probe(r, None, G, pool=True)
            feats.append(r)
        z = ops.probe_materialize(feats[4], G)
        emb = self._who(tuple(self.cid if not e else e for e in emb_ids), B, x.device)   # unet.py:186 (quirk 2: 0 means "own")
        x4, h = self.pcs_list[0]._run(z, emb)
        return feats[:4], x4, h

    def _probe_batches(self, x, G):
        """The encoder for G DIFFERENT batches (x: all G * B images) under the client's own embedding, as one batched no-grad
        train-mode pass: every batch is a statistics group -- its own BatchNorm batch statistics, its own dropout draws (the
        g-th call of each layer), the running statistics moved G times -- with the fused forms of _probe (activations
        between the convolutions never written, pooling in / before the consumer's loader).  What a decoder with autograd
        needs is then written out once per level: -> [x0, x1, x2, x3, x4 (after the channel selection)] as activations.
        Used by the ALA epoch (flower_common.MyModel._ala_epoch_batched), whose copy's encoder is frozen."""
        r = self.in_conv._probe(x, None, G)
        feats = [r]
        for blk in (self.down1, self.down2, self.down3, self.down4):
            r = blk.maxpool_conv[1]._probe(r, None, G, pool=True)
            feats.append(r)
        z = [ops.probe_materialize(f, G) for f in feats]
        who = self._who((self.cid,), x.shape[0], x.device)
        x4, _ = self.pcs_list[0]._run(z[4], who)
        return z[:4] + [x4]

    def forward(self, x, emb_idx=None):
        f, h = self._run(self._in(x), emb_idx)
        return [self._out(t) for t in f], [None if t is None else self._out(t) for t in h]


def _dsn_head(cin, n_class):
    return nn.Sequential(nn.Conv2d(cin, 512, kernel_size=3, stride=1, padding=1), nn.BatchNorm2d(512), nn.ReLU(),
                         nn.Dropout2d(0.10), nn.Conv2d(512, n_class, kernel_size=1, stride=1, padding=0, bias=False))


def _run_head(seq, x, probe=False):
    if probe:                       # output unused: only the BatchNorm statistics of the head move (see _UNetLCBase.forward)
        with torch.no_grad():       # (aux == "stats" reaches here with autograd on: the head's output does not exist, nor its graph)
            ops.conv_bn_stats_only(x.detach(), None, seq[0], seq[1], seq[3].p, "chan")
        return None
    z = ops.conv_bn_act(x, None, seq[0], seq[1], 0.0, seq[3].p, "chan")      # ReLU = slope 0; Dropout2d
    return ops.conv2d(z, None, seq[4], y_f32=True)


class _DecoderBase(_FiModule):
    def __init__(self, params):
        super().__init__()
        self.params = params
        self.in_chns, self.ft_chns = params["in_chns"], params["feature_chns"]
        self.n_class, self.bilinear = params["class_num"], params["bilinear"]
        assert len(self.ft_chns) == 5
        c = self.ft_chns
        self.up1 = UpBlock(c[4], c[3], c[3], dropout_p=0.0)
        self.up2 = UpBlock(c[3], c[2], c[2], dropout_p=0.0)
        self.up3 = UpBlock(c[2], c[1], c[1], dropout_p=0.0)
        self.up4 = UpBlock(c[1], c[0], c[0], dropout_p=0.0)
        self.out_conv = nn.Conv2d(c[0], self.n_class, kernel_size=3, padding=1)

    def _trunk(self, f, probe=False):
        x_1 = self.up1._run(f[4], f[3])
        x_2 = self.up2._run(x_1, f[2])
        x_3 = self.up3._run(x_2, f[1])
        x_4 = self.up4._run(x_3, f[0])
        # probe: out_conv has no state, its unused logits are simply not computed
        output = None if probe else ops.conv2d(x_4, None, self.out_conv, y_f32=True)    # logits always fp32
        return [output, x_1, x_2, x_3, x_4]

    def _run(self, f, probe=False, aux=True):
        return self._trunk(f, probe)

    def _heads(self):
        """(head Sequential, index of the trunk output it reads) pairs."""
        return []

    def _probe(self, skips, x4, groups):
        """Decoder of the batched no-grad forward: what it leaves behind are the BatchNorm running statistics of every block
        and head, moved `groups` times as the separate forwards would."""
        def up(blk, lo, skip, store=True):
            u = ops.probe_conv_up(lo, blk.conv1x1, groups)
            return blk.conv._probe(skip, u, groups, store=store)
        read = {idx for _, idx in self._heads()}
        o = [None, up(self.up1, x4, skips[3])]
        o.append(up(self.up2, o[1], skips[2]))
        o.append(up(self.up3, o[2], skips[1]))
        # the full-resolution output feeds the (skipped) logits convolution and, in the multi-head decoder, a head
        o.append(up(self.up4, o[3], skips[0], store=4 in read))
        for head, idx in self._heads():
            ops.probe_conv_bn(o[idx], None, head[0], head[1], 0.0, groups, store=False)

    def forward(self, feature):
        return tuple(self._out(t) for t in self._run([self._in(t) for t in feature]))


class Decoder(_DecoderBase):
    """unet.py:206-240."""


class Decoder_Head(_DecoderBase):
    """unet.py:243-285."""

    def __init__(self, params):
        super().__init__(params)
        self.dsn_head = _dsn_head(self.ft_chns[2], self.n_class)

    def _heads(self):
        return [(self.dsn_head, 2)]

    def _run(self, f, probe=False, aux=True):
        o = self._trunk(f, probe)
        return o + [_run_head(self.dsn_head, o[2], probe or aux == "stats") if aux else None]


class Decoder_MultiHead(_DecoderBase):
    """unet.py:288-346."""

    def __init__(self, params):
        super().__init__(params)
        self.dsn_head1 = _dsn_head(self.ft_chns[2], self.n_class)
        self.dsn_head2 = _dsn_head(self.ft_chns[1], self.n_class)
        self.dsn_head3 = _dsn_head(self.ft_chns[0], self.n_class)

    def _heads(self):
        return [(self.dsn_head1, 2), (self.dsn_head2, 3), (self.dsn_head3, 4)]

    def _run(self, f, probe=False, aux=True):
        o = self._trunk(f, probe)
        if not aux:
            return o + [None, None, None]
        st = probe or aux == "stats"
        return o + [_run_head(self.dsn_head1, o[2], st), _run_head(self.dsn_head2, o[3], st), _run_head(self.dsn_head3, o[4], st)]


def _params(in_chns, class_num, **extra):
    p = {"in_chns": in_chns, "feature_chns": list(FEATURE_CHNS), "dropout": list(DROPOUT), "class_num": class_num,
         "bilinear": False, "acti_func": "relu"}          # 'bilinear': False is stored but never forwarded (quirk)
    p.update(extra)
    return p


class _UNetBase(FlatStoreMixin, _FiModule):
    _decoder_cls = Decoder

    def __init__(self, in_chns, class_num):
        super().__init__()
        params = _params(in_chns, class_num)
        self.encoder = Encoder(params)
        self.decoder = self._decoder_cls(params)
        self._fi_finish_init()

    def forward(self, x):
        self._fi_refresh_packs(self.compute_dtype())     # all conv operands in one launch, only if weights changed
        f = self.encoder._run(self._in(x))
        o = self.decoder._run(f)
        return [self._out(o[0]), [self._out(t) for t in f]] + [self._out(t) for t in o[1:]]


class UNet(_UNetBase):
    """unet.py:549-566: returns [output, feature, de1, de2, de3, de4]."""


class UNet_Head(_UNetBase):
    """unet.py:640-656: + aux_output."""
    _decoder_cls = Decoder_Head


class UNet_MultiHead(_UNetBase):
    """unet.py:659-675: + aux_output1..3."""
    _decoder_cls = Decoder_MultiHead


class _UNetLCBase(FlatStoreMixin, _FiModule):
    _decoder_cls = Decoder_Head

    def __init__(self, in_chns, class_num, pcs_num, emb_num, client_num, client_id):
        super().__init__()
        params = _params(in_chns, class_num, pcs_num=pcs_num, emb_num=emb_num, client_num=client_num,
                         client_id=client_id)
        self.encoder = LCEncoder(params)
        self.decoder = self._decoder_cls(params)
        self._fi_finish_init()

    def forward(self, x, emb_idx=None, heatmap_only=False, aux=True):
        """``aux="stats"`` (train mode): the caller does not read the auxiliary outputs -- the pCE trainer uses ``[0]`` and the
        heat-map only (flower_pCE_2D.py:117-139; the heads' parameters receive no gradient there either) -- so every head runs
        its convolution as a statistics-only launch followed by the BatchNorm finalize: the running statistics move exactly as
        in the full forward (same kernel, same epilogue), the host-fed dropout draw is still taken, and the 512-channel tensor
        is never written, normalised or reduced (12 x 128^2 x 512: conv store + BN/ReLU/Dropout2d pass + 1x1 = 0.2 ms of every
        iteration).  Their entries of the returned list are None.
        ``aux=False``: the auxiliary heads are not run and their entries of the returned list are None -- for a caller that
        reads only ``[0]`` on a model whose state is thrown away afterwards (the ALA loop's deep copy, flower_common.py:503,
        566-602: nothing but the heads' BatchNorm statistics of that copy would differ).
        ``heatmap_only`` (train mode, no autograd): the caller reads nothing but the heat-map ``[6]`` -- the LC loss's
        forwards with the other clients' embeddings (flower_pCE_2D.py:128-139).  Every state change of the full forward
        still happens (all BatchNorm running statistics and counters, decoder and heads included), but tensors nobody
        reads are not produced: the heads run their convolution for its statistics only and store nothing (the full-
        resolution head alone writes, normalises and reduces an 805 MB tensor at 12x256^2), and the logits convolution,
        which has no state, is skipped.  The skipped entries of the returned list are None."""
        self._fi_refresh_packs(self.compute_dtype())     # all conv operands in one launch, only if weights changed
        probe = bool(heatmap_only) and self.training and not torch.is_grad_enabled()
        if aux == "stats" and not self.training:          # eval mode: a head nobody reads has no side effect either
            aux = False
        f, h = self.encoder._run(self._in(x), emb_idx)
        o = self.decoder._run(f, probe, aux)
        hm = [None if t is None else self._out(t) for t in h]
        out = lambda t: None if t is None else self._out(t)
        return [out(o[0]), [self._out(t) for t in f]] + [self._out(t) for t in o[1:5]] + [hm] + [out(t) for t in o[5:]]


    def probe_heatmaps(self, x, emb_ids, enc_done=None):
        """The heat-maps ``self(x, e)[6][-1]`` for every e in emb_ids -- FedICRA's LC loss asks for them once per OTHER client
        in every iteration (flower_pCE_2D.py:128-139: no-grad, train mode) -- from ONE batched pass: the K-1 forwards run
        as statistics groups of the same launches (fi_conv2d_fwd_fused / fi_bn_finalize_groups), the activations between
        the convolutions are never materialised, and the first convolution, identical under every embedding, runs once.
        State afterwards is what the K-1 separate forwards leave: every BatchNorm's running statistics moved K-1 times in
        order, num_batches_tracked += K-1, the same dropout masks.  Returns None when the batched form does not apply
        (eval mode, autograd on, host-fed masks, other architectures): the caller then makes the separate forwards."""
        enc, dec = self.encoder, self.decoder
        if not (self.training and not torch.is_grad_enabled() and ops.probe_ready() and len(emb_ids) > 0):
            return None
        if enc.n_pcs != 1 or not all(getattr(b, "bilinear", True) for b in (dec.up1, dec.up2, dec.up3, dec.up4)):
            return None
        dt = self.compute_dtype()
        vg = 4 if dt == torch.float32 else 8
        if any(c % vg for c in enc.ft_chns):
            return None
        self._fi_refresh_packs(dt)
        xin = self._in(x)
        ids = list(emb_ids)
        G, B = len(ids), xin.shape[0]
        skips, x4, h = enc._probe(xin, ids)
        maps = [self._out(h[g * B:(g + 1) * B]) for g in range(G)]
        if enc_done is not None:
            # everything the CALLER reads exists now; what follows only moves the decoder's BatchNorm statistics -- a caller that
            # runs this on a side stream may wait for this event instead of the whole stream (flower_pCE_2D._iteration)
            enc_done.record()
        dec._probe(skips, x4, G)
        return maps


class UNet_LC(_UNetLCBase):
    """unet.py:678-699: [output, feature, de1..de4, heatmap, aux_output]."""


class UNet_LC_MultiHead(_UNetLCBase):
    """unet.py:701-722: [output, feature, de1..de4, heatmap, aux_output1..3]."""
    _decoder_cls = Decoder_MultiHead
