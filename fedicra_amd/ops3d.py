"""3D companions of fedicra_amd.ops for the unet_3D surface (SURVEY.md section 8, row a18;
/root/reference/code/networks/unet_3D.py:20-94, networks/utils.py:99-123, 260-276).

Volumes are dense NDHWC tensors ``[N, D, H, W, C]`` in the compute dtype, i.e. D consecutive NHWC slices, so the 2D
kernels of libfedicra_hip.so do the heavy lifting:

* ``Conv3d(3x3x3, pad 1)`` forward / dgrad (16-bit storage, 16+ channels): ONE implicit GEMM over all slices of all volumes with
  the depth taps as channel groups of the contraction (fi_conv3d_fwd_fused / dgrad_fused); otherwise, and for wgrad
  (fi_conv3d_fwd / dgrad / wgrad): for each depth tap kd one 3x3 implicit-GEMM launch over the slices that tap reaches,
  accumulating into the output volume (the centre tap goes last: it covers every slice, so its epilogue sees the
  finished sums and produces the per-channel statistics);
* ``InstanceNorm3d(affine=False) + ReLU`` = the fused BN-finalize/apply kernel run per sample (batch statistics of
  one sample ARE its instance statistics), backward likewise through the BN backward reduce / apply kernels;
* ``MaxPool3d(2)``, trilinear x2 (align_corners False): fi_maxpool3d_*, fi_upsample3d2x_*;
* ``Dropout``: the counter-RNG mask of the 2D path (regenerated in backward).

No CPU fallback: everything routes through fedicra_amd._lib.
"""
from __future__ import annotations

import os

import torch
from torch.autograd import Function

from . import _lib as L
from . import ops


_pack_cache = {}               # id(weight) -> (weakref, key, {what: operand}): the packed operands of the current weights


def _cached(weight, dtype, what, build):
    """Packed operands are rebuilt when the weights change -- in place (``_version``) or through raw pointers (fused AdamW / ALA
    kernels / graph replay: ops.weights_epoch) -- not at every forward and backward call."""
    import weakref
    key = (weight._version, ops.weights_epoch(), dtype, weight.data_ptr(), torch.cuda.is_current_stream_capturing())
    ent = _pack_cache.get(id(weight))
    if ent is None or ent[0]() is not weight or ent[1] != key:
        if len(_pack_cache) > 4096:
            _pack_cache.clear()
        ent = (weakref.ref(weight), key, {})
        _pack_cache[id(weight)] = ent
    if what not in ent[2]:
        ent[2][what] = build()
    return ent[2][what]


def _w_taps(weight, dtype, mode):
    """Conv3d weight [Cout,Cin,kD,kH,kW] -> per depth tap the packed 2D operand ([Cout][kH*kW][Cin] forward,
    flipped/transposed for dgrad)."""
    def build():
        cout, cin, kd, kh, kw = weight.shape
        out = []
        for t in range(kd):
            wk = weight[:, :, t].permute(0, 2, 3, 1).contiguous().float()          # [Cout,kH,kW,Cin] fp32
            dst = torch.empty(cout * kh * kw * cin, dtype=dtype, device=weight.device)
            L.pack_weights(wk, dst, cout, kh * kw, cin, mode)
            out.append(dst)
        return out
    return _cached(weight, dtype, ("taps", mode), build)


class _MultiPack3D:
    """Every eligible Conv3d weight that asked for a one-launch operand so far, their persistent operand buffers and the device
    table fi_pack_weights3d_multi reads: ONE launch rebuilds both operands of all of them when any went stale (the optimizer step
    of an iteration makes all of them stale at once) instead of three cast / flip / permute launches per layer."""

    def __init__(self, dtype):
        self.dtype = dtype
        self.ents = []              # [weakref(weight), forward operand, dgrad operand, the parameter's address in the table]
        self.index = {}             # id(weight) -> position
        self.table = None
        self.nblocks = 0

    @staticmethod
    def eligible(weight):
        if weight.dim() != 5 or not weight.is_cuda or weight.dtype != torch.float32 or not weight.is_contiguous():
            return False
        cout, cin, kd, kh, kw = weight.shape
        return kd == kh == kw == 3 and cin >= 8 and cout >= 8

    def ensure(self, weight):
        """True when `weight` and everything else registered can be packed by the table as it stands; rebuilds the table (dead
        weights dropped, moved ones re-addressed, `weight` added) when not -- except inside a stream capture, where a host -> device
        copy is not allowed: False then, and the caller builds this operand the per-layer way."""
        import weakref
        pos = self.index.get(id(weight))
        fresh = (pos is not None and self.ents[pos][0]() is weight
                 and all(e[0]() is not None and e[0]().data_ptr() == e[3] for e in self.ents))
        if fresh:
            return True
        if torch.cuda.is_current_stream_capturing():
            return False
        live = [e for e in self.ents if e[0]() is not None and e[0]().device == weight.device]
        if not any(e[0]() is weight for e in live):
            cout, cin = weight.shape[0], weight.shape[1]
            live.append([weakref.ref(weight), torch.empty((cout, 9, 3, cin), dtype=self.dtype, device=weight.device),
                         torch.empty((cin, 9, 3, cout), dtype=self.dtype, device=weight.device), 0])
        rows, nb = [], 0
        for e in live:
            w = e[0]()
            e[3] = w.data_ptr()
            rows.append([e[3], e[1].data_ptr(), e[2].data_ptr(), w.shape[0], w.shape[1], nb])
            nb += -(-(w.shape[0] * w.shape[1]) // 256)
        self.ents = live
        self.index = {id(e[0]()): i for i, e in enumerate(live)}
        self.table = torch.tensor(rows, dtype=torch.int64).to(weight.device)
        self.nblocks = nb
        return True

    def refresh(self):
        """Repack everything registered and install the operands in the pack cache under the keys _cached() looks up now."""
        import weakref
        L.pack_weights3d_multi(self.table, len(self.ents), self.nblocks, self.dtype)
        cap = torch.cuda.is_current_stream_capturing()
        for wr, d0, d1, _ in self.ents:
            w = wr()
            key = (w._version, ops.weights_epoch(), self.dtype, w.data_ptr(), cap)
            ent = _pack_cache.get(id(w))
            if ent is None or ent[0]() is not w or ent[1] != key:
                ent = (weakref.ref(w), key, {})
                _pack_cache[id(w)] = ent
            ent[2][("all", 0)] = d0
            ent[2][("all", 1)] = d1


_multi3d = {}                  # dtype -> _MultiPack3D
_MULTI3D = __import__("os").environ.get("FI_PACK3D_MULTI", "1") != "0"      # measurement switch: 0 = per-layer torch copies


def _w_all(weight, dtype, mode):
    """The one-launch forms' operands straight from the parameter, ONE cast-and-permute copy each (two for the dgrad side: the flip
    materialises) instead of per depth tap a slice copy, fi_pack_weights and a stack -- 16 small launches per layer and iteration
    that were a ninth of a unet_3D step (profiles/r04_z_c4_bench_kernel_stats.csv: 107 pack + 123 copy + 34 cat launches).
    mode 0: [Cout][kH*kW][kD][Cin] = weight[co][ci][t][r][s] (fi_pack_weights mode 0 per tap, stacked over the taps);
    mode 1: [Cin][kH*kW][kD][Cout] with all three filter axes reversed (mode 1 per tap, the taps stacked in reverse).
    Values: the same round-to-nearest cast fi_pack_weights makes."""
    def build():
        cout, cin, kd, kh, kw = weight.shape
        w = weight.detach()
        if mode == 0:
            out = torch.empty((cout, kh * kw, kd, cin), dtype=dtype, device=weight.device)
            out.view(cout, kh, kw, kd, cin).copy_(w.permute(0, 3, 4, 2, 1))
        else:
            out = torch.empty((cin, kh * kw, kd, cout), dtype=dtype, device=weight.device)
            out.view(cin, kh, kw, kd, cout).copy_(w.flip(2, 3, 4).permute(1, 3, 4, 2, 0))
        return out
    if _MULTI3D and dtype != torch.float32 and _MultiPack3D.eligible(weight):
        key = (weight._version, ops.weights_epoch(), dtype, weight.data_ptr(), torch.cuda.is_current_stream_capturing())
        ent = _pack_cache.get(id(weight))
        if ent is None or ent[0]() is not weight or ent[1] != key or ("all", mode) not in ent[2]:
            mp = _multi3d.get(dtype)
            if mp is None:
                mp = _multi3d[dtype] = _MultiPack3D(dtype)
            if mp.ensure(weight):
                mp.refresh()                         # this weight is stale: so is every other one -- all of them in one launch
    return _cached(weight, dtype, ("all", mode), build)


_const_cache = {}
_capture_consts = {}          # constants first made inside the CURRENT stream capture: (device, width) -> tensors
_capture_state = [False]      # was the previous call inside a capture?  (a capture that begins drops the previous capture's set)


def _norm_consts(cout, dev):
    """(ones, zeros, running-mean scratch, running-var scratch) of an affine-free InstanceNorm: the same four vectors for every
    layer of that width (momentum 0 leaves the scratch statistics as they are) -- four fill launches per convolution otherwise.
    Cached per (device, width) process-wide only when created EAGERLY: tensors first made inside a stream capture live in that
    graph's private pool and their fills are merely recorded, so another graph must not find them.  A capture
    without an eager warm-up keeps its own set for ITS duration: one set per width and capture, not four fills per layer baked
    into the graph."""
    key = (dev.index, cout)
    c = _const_cache.get(key)
    if c is not None:
        return c
    capturing = torch.cuda.is_current_stream_capturing()
    if capturing and not _capture_state[0]:
        _capture_consts.clear()                      # a new capture began
    _capture_state[0] = capturing
    if capturing:
        c = _capture_consts.get(key)
        if c is not None:
            return c
    c = (torch.ones(cout, device=dev), torch.zeros(cout, device=dev), torch.zeros(cout, device=dev),
         torch.ones(cout, device=dev))
    if capturing:
        _capture_consts[key] = c
    else:
        _const_cache[key] = c
    return c


_IN_BATCHED = os.environ.get("FI_IN_BATCHED", "1") != "0"      # measurement switch: 0 = one InstanceNorm launch per sample and pass
_WGRAD3D_DEFER = os.environ.get("FI_WGRAD3D_DEFER", "1") != "0"   # measurement switch: 0 = a reduce launch + a permuted add per 3x3x3 layer
_POOLSKIP3D = os.environ.get("FI_POOLSKIP3D", "1") != "0"        # measurement switch: 0 = autograd adds skip and pooling gradients itself
_FIRST3D = os.environ.get("FI_FIRST3D", "1") != "0"            # measurement switch: 0 = the per-tap implicit GEMMs of rounds 1-4


def _first_ok(dt, ydt, kd, ksize, x0, x1, cout):
    """The shape csrc/conv3d_first.hip covers: ONE input channel, 16 output channels, 3x3x3, 16-bit storage."""
    return (_FIRST3D and dt != torch.float32 and ydt == dt and kd == 3 and ksize == 3 and x1 is None and x0.shape[4] == 1
            and cout == 16 and x0.numel() * 32 < (1 << 31) - 64)      # (byte offsets into the 16-channel side stay below 2^31: the kernel marks
                                                                        #  out-of-image lanes by setting bit 31 of the offset -- ADVICE r5)


def _point_ok(dt, ydt, kd, ksize, x0, x1, cout, norm):
    """The shape csrc/conv3d_point.hip covers: 16 channels in, <= 4 fp32 channels out, 1x1x1, no normalisation, 16-bit input."""
    return (_FIRST3D and not norm and dt != torch.float32 and ydt == torch.float32 and kd == 1 and ksize == 1 and x1 is None
            and x0.shape[4] == 16 and cout <= 4)


def _w_point(weight):
    """[Cout, 16, 1, 1, 1] (any strides) -> dense fp32 [Cout, 16]."""
    return weight.detach().reshape(weight.shape[0], 16).float().contiguous()


def _w27(weight):
    """[16, 1, 3, 3, 3] (any strides) -> dense fp32 [16, 27], taps in (kd, kh, kw) order."""
    return weight.detach().reshape(weight.shape[0], 27).float().contiguous()


def _fused_ok(dt, ydt, kd, ksize, x0, x1, cout):
    """Shapes fi_conv3d_*_fused covers (the library re-checks and answers FI_ERR_UNSUPPORTED otherwise)."""
    if dt == torch.float32 or ydt != dt or kd != 3 or ksize != 3:
        return False
    c0 = x0.shape[4]
    c1 = 0 if x1 is None else x1.shape[4]
    return c0 % 8 == 0 and c1 % 8 == 0 and c0 + c1 >= 16 and cout >= 16 and cout % 8 == 0 and x0.shape[2] >= 8


class _Conv3d(Function):
    """y = conv3d(cat(x0, x1)) + bias, optionally followed by InstanceNorm3d(affine=False) + ReLU."""

    @staticmethod
    def forward(ctx, x0, x1, weight, bias, norm, y_f32, mod=None):
        ctx.mod = mod
        N, D, H, W, _ = x0.shape
        cout, cin, kd, ksize, _ = weight.shape
        dev, dt = x0.device, x0.dtype
        ydt = torch.float32 if (y_f32 and not norm) else dt
        # per-sample statistics accumulators out of the iteration's zeroed arena (one memset per iteration, not a fill per layer)
        nst = L.STATS_SLOTS * cout * 2
        stats = ops._ctx.arena.take(N * nst, dev).view(N, nst) if norm else None
        x0c, x1c = x0.contiguous(), None if x1 is None else x1.contiguous()
        y = None
        if _point_ok(dt, ydt, kd, ksize, x0c, x1c, cout, norm):
            # the network's last convolution, 16 -> n_classes at 1x1x1 to fp32 logits: one streaming pass (csrc/conv3d_point.hip)
            # instead of a zero fill + an accumulating padded GEMM per sample
            y = torch.empty((N, D, H, W, cout), dtype=torch.float32, device=dev)
            L.conv3d_point_fwd(x0c, _w_point(weight), bias, y)
        elif _first_ok(dt, ydt, kd, ksize, x0c, x1c, cout):
            # the network's first convolution, 1 -> 16 channels: one pass on the matrix pipe that streams the output once
            # (csrc/conv3d_first.hip) instead of three read-modify-write passes of a K = 27 (padded to 32) implicit GEMM
            y = torch.empty((N, D, H, W, cout), dtype=dt, device=dev)
            L.conv3d_first_fwd(x0c, _w27(weight), bias, y, stats)
        elif _fused_ok(dt, ydt, kd, ksize, x0c, x1c, cout):
            # ONE implicit GEMM over all slices of all volumes, the depth taps as channel groups of its contraction: y is
            # written once (the per-tap form below reads and rewrites it twice more)
            y = torch.empty((N, D, H, W, cout), dtype=dt, device=dev)
            if not L.conv3d_fwd_fused(x0c, x1c, _w_all(weight, dt, 0), bias, y, stats, ksize=ksize):      # [Cout][9][3][Cin]
                y = None
        if y is None:
            y = torch.zeros((N, D, H, W, cout), dtype=ydt, device=dev)
            # every (sample, depth tap) 2D launch is issued by ONE C-ABI call (fi_conv3d_fwd)
            L.conv3d_fwd(x0c, x1c, _w_taps(weight, dt, 0), bias, y, stats, ksize=ksize,
                         y_f32=ydt == torch.float32 and dt != torch.float32)
        if not norm:
            ctx.save_for_backward(x0, x1, weight)
            ctx.norm, ctx.has_bias = False, bias is not None
            return y
        z = torch.empty_like(y)
        coef = torch.empty((N, 4, cout), dtype=torch.float32, device=dev)
        one, zero, rm, rv = _norm_consts(cout, dev)                 # (rm, rv: scratch -- no running statistics)
        if _IN_BATCHED and N > 1 and y.is_contiguous():
            # every sample of the batch in ONE launch (blockIdx.y = sample: its own statistics and coefficient rows)
            L.instnorm_fwd_batched(y, z, stats, one, zero, rm, rv, 1e-5, coef)
        else:
            for n in range(N):
                L.bn_fused_fwd(y[n], z[n], stats[n], one, zero, rm, rv, None, 0.0, 1e-5, True, coef[n], 0.0, None)
        ctx.save_for_backward(x0, x1, weight, y, coef)
        ctx.norm, ctx.has_bias = True, bias is not None
        return z

    @staticmethod
    def backward(ctx, dz):
        if ctx.norm:
            x0, x1, weight, y, coef = ctx.saved_tensors
        else:
            x0, x1, weight = ctx.saved_tensors
        N, D, H, W, c0 = x0.shape
        cout, cin, kd, ksize, _ = weight.shape
        dev, dt = x0.device, x0.dtype
        dz = dz.contiguous()
        if ctx.norm:
            dy = torch.empty_like(y)
            if _IN_BATCHED and N > 1 and y.is_contiguous():
                nst = L.STATS_SLOTS * cout * 2
                sums = ops._ctx.arena.take(N * nst, dev).view(N, nst)
                L.instnorm_bwd_batched(dz, y, coef, sums, dy)       # two launches for the whole batch (reduce, apply)
            else:
                for n in range(N):
                    sums = ops._ctx.arena.take(L.STATS_SLOTS * cout * 2, dev)
                    L.bn_act_bwd_reduce(dz[n], y[n], coef[n, 0], coef[n, 1], coef[n, 2], coef[n, 3], sums, 0.0, None)
                    L.bn_act_bwd_apply(dz[n], y[n], coef[n, 0], coef[n, 1], coef[n, 2], coef[n, 3], sums, True, dy[n], None,
                                       None, 0.0, None)
        elif _point_ok(dt, dz.dtype, kd, ksize, x0, x1, cout, False):
            # (csrc/conv3d_point.hip: the fp32 logit gradient is consumed as it is -- no cast pass -- by one streaming launch per
            #  direction; gradient targets as below)
            x0c = x0.contiguous()
            w2 = _w_point(weight)
            dx0 = gw = gb = None
            if ctx.needs_input_grad[0]:
                dx0 = torch.empty_like(x0c)
                L.conv3d_point_dgrad(dz, w2, dx0)
            need_w, need_b = ctx.needs_input_grad[2], ctx.has_bias and ctx.needs_input_grad[3]
            if need_w or need_b:
                mod = ctx.mod
                wt_ = gw = None
                if need_w:
                    if mod is not None:
                        wt_, _, gw = ops._grad_target(mod.weight)
                    else:
                        wt_ = gw = torch.zeros_like(weight, dtype=torch.float32)
                db = None
                if need_b:
                    if mod is not None:
                        db, _, gb = ops._grad_target(mod.bias)
                    else:
                        db = gb = torch.zeros(cout, dtype=torch.float32, device=dev)
                g = None
                if need_w:
                    g = ops._ctx.arena.take((cout * 16 + 1) // 2, dev).view(torch.float32)[:cout * 16]     # zeros of the iteration's arena
                L.conv3d_point_wgrad(x0c, dz, g, db)
                if need_w:
                    wt_.add_(g.view(cout, 16, 1, 1, 1))
            return dx0, None, gw, gb, None, None, None
        else:
            dy = dz if dz.dtype == dt else dz.to(dt)
        need_x0, need_x1 = ctx.needs_input_grad[0], x1 is not None and ctx.needs_input_grad[1]
        dx0 = dx1 = gw = gb = None
        if need_x0 or need_x1:
            dyc = dy.contiguous()
            done = False
            if _fused_ok(dt, dt, kd, ksize, dyc, None, c0) and (x1 is None or x1.shape[4] % 8 == 0):
                d0 = torch.empty_like(x0)
                d1 = None if x1 is None else torch.empty_like(x1)
                # the depth taps reversed: input slice d + j - 1 of dy meets the filter's depth tap 2 - j
                done = L.conv3d_dgrad_fused(dyc, _w_all(weight, dt, 1), d0, d1, ksize=ksize)
            if not done:
                d0 = torch.zeros_like(x0)
                d1 = None if x1 is None else torch.zeros_like(x1)
                L.conv3d_dgrad(dyc, _w_taps(weight, dt, 1), d0, d1, ksize=ksize)
            dx0 = d0 if need_x0 else None
            dx1 = d1 if need_x1 else None
        if ctx.needs_input_grad[2] or (ctx.has_bias and ctx.needs_input_grad[3]):
            # Gradient targets like the 2D path's (ops._grad_target): a parameter of a flat-store model accumulates straight into
            # its slice of the flat gradient buffer -- the bias through the kernel's own `+=`, the filter through ONE add that reads
            # the kernel's [Cout][9][3][Cin] layout -- and autograd gets None; a fresh gradient tensor, a fill, a permuted copy and
            # the optimizer's gather copy per parameter otherwise (40 device copies + 19 fills per unet_3D iteration).
            mod = ctx.mod
            need_w, need_b = ctx.needs_input_grad[2], ctx.has_bias and ctx.needs_input_grad[3]
            if need_w and mod is not None:
                wt_, _, gw = ops._grad_target(mod.weight)
            else:                                   # (a frozen filter must not be handed a gradient: scratch, dropped)
                wt_ = torch.zeros_like(weight, dtype=torch.float32)
                gw = wt_ if need_w else None
            db = gb = None
            if ctx.has_bias:
                if need_b and mod is not None:
                    db, _, gb = ops._grad_target(mod.bias)
                else:
                    db = torch.zeros(cout, dtype=torch.float32, device=dev)
                    gb = db if need_b else None
            x0c, x1c, dyc = x0.contiguous(), None if x1 is None else x1.contiguous(), dy.contiguous()
            done = False
            if _first_ok(dt, dyc.dtype, kd, ksize, x0c, x1c, cout):
                # (csrc/conv3d_first.hip: the stencil's mirror image, one partial slice per workgroup + a fixed-order sum)
                g27 = ops._ctx.arena.take((cout * 27 + 1) // 2, dev).view(torch.float32)[:cout * 27]      # zeros of the iteration's arena
                L.conv3d_first_wgrad(x0c, dyc, g27, db)
                wt_.add_(g27.view(cout, 1, 3, 3, 3))
                done = True
            if not done and kd == 3 and ksize == 3:
                # ONE launch over all slices of all volumes, the depth taps as channel groups of the input side
                # (zeros out of the iteration's arena -- fp64 words viewed as fp32; the tensor dies with the add below)
                if _WGRAD3D_DEFER and mod is not None and need_w and gw is None and (db is None or gb is None) and wt_.is_contiguous():
                    # both targets are gradient SINKS (a flat-store model): stage 1 now, stage 2 of ALL layers as one launch at
                    # the end of the backward pass, written straight into the parameter layout (ops.flush_wgrad, table word 9) --
                    # a reduce launch and a permuted add per layer otherwise (17 + 17 launches, 0.28 ms of a unet_3D iteration)
                    part = L.conv3d_wgrad_fused_partial(x0c, x1c, dyc, db is not None, ksize=ksize)
                    if part is not None:
                        ws, slices, stride = part
                        ops._defer_wgrad_reduce(ws, stride, slices, wt_, cout * 27 * cin, db, cout, None, cin3=cin)
                        done = True
            if not done and kd == 3 and ksize == 3:
                ng = cout * ksize * ksize * kd * cin
                gall = ops._ctx.arena.take((ng + 1) // 2, dev).view(torch.float32)[:ng].view(cout, ksize, ksize, kd, cin)   # [Cout][9][3][Cin]
                if L.conv3d_wgrad_fused(x0c, x1c, dyc, gall, db, ksize=ksize):        # (declines before it launches anything)
                    wt_.add_(gall.permute(0, 4, 3, 1, 2))              # -> [Cout,Cin,kD,kH,kW]
                    done = True
            if not done:
                gwk = torch.zeros((kd, cout, ksize, ksize, cin), dtype=torch.float32, device=dev)
                L.conv3d_wgrad(x0c, x1c, dyc, gwk, db, ksize=ksize)
                wt_.add_(gwk.permute(1, 4, 0, 2, 3))                   # -> [Cout,Cin,kD,kH,kW]
        return dx0, dx1, gw, gb, None, None, None


class _PoolSkip3d(Function):
    """(skip, pooled) = (z, maxpool3d(z)) for an encoder feature that feeds both the next level and a decoder skip connection
    (/root/reference/code/networks/unet_3D.py:63-76: conv1..conv4 are pooled AND concatenated in the up path).  autograd would add
    the two gradients with an elementwise launch of its own (4 per iteration, 84 us at 2 x 128^3); backward here makes the sum in
    the pooling pass (ops._PoolSkip's 3D sibling; even extents only -- the caller falls back to maxpool3d otherwise)."""

    @staticmethod
    def forward(ctx, x):
        N, D, H, W, Cc = x.shape
        y = torch.empty((N, D // 2, H // 2, W // 2, Cc), dtype=x.dtype, device=x.device)
        L.maxpool3d_fwd(x, y)
        ctx.save_for_backward(x)
        return x.view_as(x), y

    @staticmethod
    def backward(ctx, dskip, dy):
        (x,) = ctx.saved_tensors
        if dy is None:
            return dskip
        dx = torch.empty_like(x)
        if dskip is None:
            L.maxpool3d_bwd(x, dy.contiguous(), dx)
        else:
            if dskip.dtype != x.dtype:
                dskip = dskip.to(x.dtype)
            L.maxpool3d_bwd_add(x, dy.contiguous(), dskip.contiguous(), dx)
        return dx


class _MaxPool3d(Function):
    """nn.MaxPool3d(2): windows that do not fit are dropped (floor), i.e. an odd extent loses its last plane."""

    @staticmethod
    def forward(ctx, x):
        N, D, H, W, Cc = x.shape
        ctx.full = None
        if (D | H | W) & 1:
            ctx.full = x.shape
            x = x[:, :D // 2 * 2, :H // 2 * 2, :W // 2 * 2].contiguous()
        y = torch.empty((N, D // 2, H // 2, W // 2, Cc), dtype=x.dtype, device=x.device)
        L.maxpool3d_fwd(x, y)
        ctx.save_for_backward(x)
        return y

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        dx = torch.empty_like(x)
        L.maxpool3d_bwd(x, dy.contiguous(), dx)
        if ctx.full is not None:                        # the dropped planes receive no gradient
            out = torch.zeros(ctx.full, dtype=dx.dtype, device=dx.device)
            out[:, :x.shape[1], :x.shape[2], :x.shape[3]] = dx
            dx = out
        return dx


class _Upsample3d(Function):
    @staticmethod
    def forward(ctx, x):
        N, d, h, w, Cc = x.shape
        y = torch.empty((N, 2 * d, 2 * h, 2 * w, Cc), dtype=x.dtype, device=x.device)
        L.upsample3d2x_fwd(x, y)
        ctx.shape = x.shape
        return y

    @staticmethod
    def backward(ctx, dy):
        dx = torch.empty(ctx.shape, dtype=dy.dtype, device=dy.device)
        L.upsample3d2x_bwd(dy.contiguous(), dx)
        return dx


class _Dropout(Function):
    """Element-wise dropout on a dense channels-last tensor: z = x * mask / (1 - p); backward applies the same
    (regenerated) mask to the gradient."""

    @staticmethod
    def forward(ctx, x, p, owner, channel=False):
        C = x.shape[-1]
        # channel = True: Dropout3d, one draw per (sample, channel) -- the per-sample view gives the kernel its sample index
        v = x.reshape(x.shape[0], -1, 1, C) if channel else x.reshape(-1, 1, 1, C)
        spec = ops._drop_spec(p, "chan" if channel else "elem", v.shape[0], v.shape[1], 1, C, x.device, owner=owner)
        one, zero = _norm_consts(C, x.device)[:2]
        z = torch.empty_like(v)
        L.bn_act_fwd(v, one, zero, z, 1.0, spec)
        ctx.spec, ctx.C, ctx.vshape = spec, C, tuple(v.shape)
        return z.reshape(x.shape)

    @staticmethod
    def backward(ctx, dz):
        C = ctx.C
        v = dz.contiguous().reshape(ctx.vshape)
        one, zero = _norm_consts(C, dz.device)[:2]
        out = torch.empty_like(v)
        L.bn_act_fwd(v, one, zero, out, 1.0, ctx.spec)
        return out.reshape(dz.shape), None, None, None


def conv3d(x0, x1, conv, norm=False, y_f32=False):
    """conv: nn.Conv3d (kernel 3 pad 1, or kernel 1).  x1: second input of a channel concatenation or None."""
    k = conv.kernel_size
    if not (k[0] == k[1] == k[2] and k[0] in (1, 3) and conv.padding == (k[0] // 2,) * 3 and conv.stride == (1, 1, 1)):
        raise NotImplementedError("conv3d: only 3x3x3/pad 1 and 1x1x1, stride 1 (all the unet_3D surface uses)")
    return _Conv3d.apply(x0, x1, conv.weight, conv.bias, bool(norm), bool(y_f32), conv)


def maxpool3d(x):
    return _MaxPool3d.apply(x)


def pool_skip3d(x):
    """Returns (x, maxpool3d(x)); use the first result wherever x is consumed besides the pooling (FI_POOLSKIP3D=0, odd extents or a
    tensor that needs no gradient: the plain pooling)."""
    if _POOLSKIP3D and x.requires_grad and not ((x.shape[1] | x.shape[2] | x.shape[3]) & 1):
        return _PoolSkip3d.apply(x)
    return x, _MaxPool3d.apply(x)


def upsample3d2x(x):
    return _Upsample3d.apply(x)


def dropout(x, p, training, owner=None, channel=False):
    if not training or p <= 0.0:
        return x
    return _Dropout.apply(x, float(p), owner, bool(channel))
