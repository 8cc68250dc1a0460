"""torch.autograd.Function wrappers over the C ABI (include/fedicra_hip.h).

Internal tensor convention: activations are dense NHWC tensors ``[N,H,W,C]`` in the compute
dtype (float32 = exact-fp32 parity mode, bfloat16 = performance mode); parameters are always
fp32, conv weights logically ``[Cout,Cin,k,k]`` with channels_last strides (memory
``[Cout][k][k][Cin]``).  The nn.Modules in fedicra_amd/networks present the reference's NCHW
surface as zero-copy permuted views.

Weight gradients use a *grad sink*: parameters created by ``FlatStore`` carry ``_fi_gview`` (a
view into one flat fp32 gradient buffer).  backward() accumulates straight into it (the wgrad
kernel adds atomically, i.e. autograd's ``+=`` semantics), sets ``p.grad`` to that view and
returns None for the parameter, so that the fused AdamW and the RCCL aggregation run over
flat buffers.  Parameters without a sink get an ordinary returned gradient.
"""
from __future__ import annotations

import contextlib
import itertools
import os

import torch
from torch.autograd import Function

from . import _lib as L

# ----------------------------------------------------------------------------- dropout control
_drop_seed = itertools.count(0x5EED)
_mask_provider = None          # parity mode: callable(shape_nchw, p) -> uint8/bool/float keep mask (NCHW, any device)


def set_dropout_mask_provider(fn):
    """Parity mode: dropout masks come from ``fn(shape_nchw, p)`` instead of the device RNG."""
    global _mask_provider
    _mask_provider = fn


def set_dropout_seed_offset(t):
    """Device int32[1] mixed into every dropout seed; lets a replayed hipGraph draw fresh masks."""
    _ctx.seed_offset = t


def manual_seed(seed: int):
    global _drop_seed, _base_seed
    _drop_seed = itertools.count(int(seed) * 0x9E3779B1 + 0x5EED)
    _base_seed = int(seed) * 0x9E3779B1 + 0x5EED


_base_seed = 0x5EED
_layer_uid = {}                # id(bn module) -> small stable integer (order of first use)


class _ZeroArena:
    """fp64 scratch for the BatchNorm statistic / gradient-sum accumulators of one iteration.  The kernels add
    into zero-initialised buffers; handing those out of ONE arena that begin_iteration() clears with a single
    memset replaces ~37 `torch.zeros` fill launches per training step.  take() never returns dirty memory: the
    offset only rewinds together with the memset, and requests beyond the arena fall back to torch.zeros."""

    SIZE = 1 << 22          # doubles (32 MB)

    def __init__(self):
        self.buf = None
        self.off = 0
        self.hwm = 0

    def begin(self, device):
        if self.buf is None or self.buf.device != device:
            self.buf = torch.zeros(self.SIZE, dtype=torch.float64, device=device)
            self.off = self.hwm = 0
            return
        n = max(self.hwm, self.off)
        self.hwm = n
        if n:
            self.buf[:n].zero_()
        self.off = 0

    def take(self, n, device):
        n = (n + 15) // 16 * 16
        if self.buf is not None and self.buf.device == device and self.off + n <= self.SIZE:
            v = self.buf[self.off:self.off + n]
            self.off += n
            return v
        return torch.zeros(n, dtype=torch.float64, device=device)


class _Context:
    """The device-side state one training client owns and a captured hipGraph of its step bakes in by address: the
    accumulator arena, the iteration counter mixed into the dropout seeds, the per-iteration mask call counts.  Several
    clients hosted by one process (one HIP stream each, flower_pCE_2D.MyClient.train_steps) must not share it -- two
    graphs replaying concurrently would add into the same accumulators.  Single host thread: the current context is a
    module global switched by use_context()."""

    def __init__(self):
        self.arena = _ZeroArena()
        self.seed_offset = None        # device int32[1] added to every RNG seed (training-iteration counter)
        self.call_idx = {}             # uid -> how many times this layer drew a mask in the current iteration
        self.graph_tables = []         # pinned + device reduce tables of the steps captured under this context
        # list while they do (flower_pCE_2D._iteration): the running-statistics updates of the batched forward, which the
        # CALLER makes on its own stream after joining the probe stream (_probe_finalize); None = made in place
        self.probe_deferred = None


_ctx = _Context()


def new_context():
    return _Context()


@contextlib.contextmanager
def use_context(ctx):
    global _ctx
    prev, _ctx = _ctx, ctx
    try:
        yield ctx
    finally:
        _ctx = prev


_raw_epoch = 0                 # bumped whenever a kernel writes model weights through raw pointers


def bump_weights_epoch():
    """Tell the pack cache that weights changed behind torch's back (fused AdamW, ALA kernel, graph replay)."""
    global _raw_epoch
    _raw_epoch += 1


def weights_epoch():
    return _raw_epoch


def begin_iteration(device=None):
    """Call at the start of every training iteration: dropout seeds are then a pure function of
    (manual_seed, layer, n-th call of that layer within the iteration, device iteration counter), so an
    eager run and a replayed hipGraph draw identical masks, while repeated forwards inside one iteration
    (FedICRA's no-grad forwards with other clients' embeddings) still get independent masks.  Also rewinds
    and clears the accumulator arena (one memset)."""
    _ctx.call_idx.clear()
    L.profile_block()
    if device is not None:
        _ctx.arena.begin(torch.device(device))


def _drop_spec(p, kind, N, H, W, Cc, device, owner=None):
    """kind: 'elem' (nn.Dropout) or 'chan' (nn.Dropout2d).  Returns the tuple _lib._bnact expects."""
    if p <= 0.0:
        return None
    if _mask_provider is not None:
        shape = (N, Cc, H, W) if kind == "elem" else (N, Cc, 1, 1)
        m = _mask_provider(shape, p)
        m = (m != 0).to(torch.uint8)
        m = m.permute(0, 2, 3, 1).contiguous().to(device) if kind == "elem" else m.reshape(N, Cc).contiguous().to(device)
        return (L.DROP_MASK_ELEM if kind == "elem" else L.DROP_MASK_CHAN, float(p), 0, m, None)
    mode = L.DROP_RNG_ELEM if kind == "elem" else L.DROP_RNG_CHAN
    if _ctx.seed_offset is None or owner is None:
        return (mode, float(p), next(_drop_seed), None, _ctx.seed_offset)
    uid = getattr(owner, "_fi_uid", None)         # FlatStoreMixin numbers the BN modules of a model in order
    if uid is None:
        uid = _layer_uid.setdefault(id(owner), 1000 + len(_layer_uid))
    k = _ctx.call_idx.get(uid, 0)
    _ctx.call_idx[uid] = k + 1
    return (mode, float(p), _base_seed + (uid << 24) + k * 0x10001, None, _ctx.seed_offset)


# ----------------------------------------------------------------------------- helpers
def _krsc(w):
    """[Cout,Cin,k,k] parameter -> dense [Cout,k,k,Cin] view (copy only if not channels_last)."""
    v = w.permute(0, 2, 3, 1)
    return v if v.is_contiguous() else v.contiguous()


def _grad_target(p, like=None):
    """Returns (dense fp32 tensor to accumulate into, fresh: bool, returned_grad or None)."""
    sink = getattr(p, "_fi_gview", None)
    if sink is None:
        g = torch.zeros_like(p if like is None else like, dtype=torch.float32, memory_format=torch.preserve_format)
        return g, True, g
    if p.grad is None:
        if not getattr(p, "_fi_zeroed", False):     # FlatStoreMixin.zero_grad() zeroes all sinks in one memset
            sink.zero_()
        p._fi_zeroed = False
        p.grad = sink
        return sink, True, None
    if p.grad.data_ptr() != sink.data_ptr():
        g = torch.zeros_like(p, dtype=torch.float32, memory_format=torch.preserve_format)
        return g, True, g
    return sink, False, None


def _packed(wk, dtype, mode, cout, kk, cin, param=None):
    """fp32 KRSC weight -> operand in `dtype` (mode 0) or flipped/transposed dgrad operand (mode 1).
    Parameters owned by a FlatStoreMixin model are served from the model's multi-tensor pack buffers when those
    are current (the root module's forward refreshes all of them in ONE launch); otherwise a per-tensor repack."""
    if mode == 0 and dtype == torch.float32:
        return wk
    if param is not None:
        packs = getattr(param, "_fi_packs", None)
        if packs is not None:
            owner = packs[0]()
            if owner is not None and owner._fi_packs_current(dtype):
                return packs[1 + mode]
    out = torch.empty(cout * kk * cin, dtype=dtype, device=wk.device)
    L.pack_weights(wk, out, cout, kk, cin, mode)
    ksize = 3 if kk == 9 else (1 if kk == 1 else 0)
    if wk.is_cuda and L.conv_weight_chunk16(dtype, ksize, *((cin, cout) if mode == 0 else (cout, cin))):
        # the chunk-major sibling the 64 x 64-wave-tile kernel stages from (include/fedicra_hip.h: FiConv.w16)
        out16 = torch.empty_like(out)
        L.pack_weights(wk, out16, cout, kk, cin, 2 + mode)
        out._fi_w16 = out16
    return out


_pending_wgrad = []            # (workspace, stride, slices, dw, n_dw, dbias, cout) awaiting the multi-tensor reduce
_wgrad_cb_queued = False
_table_keepalive = []          # pinned host tables of eager launches: must outlive the async copy


def stream_beside(cached, main, device=None, priority=0):
    """A stream that is NOT `main`, for work meant to run beside it.  torch hands out streams from a round-robin pool of
    32 per device and priority: in a long-lived process a freshly made stream can BE the one an earlier caller holds --
    e.g. torch.cuda.graph's capture stream -- and a branch "forked" onto the stream it came from is silently serial.
    `cached` (or None) is kept when it differs from `main`; otherwise a new one is drawn until it does."""
    if cached is not None and cached.cuda_stream != main.cuda_stream:
        return cached
    for _ in range(64):
        s = torch.cuda.Stream(device=device, priority=priority)
        if s.cuda_stream != main.cuda_stream:
            return s
    raise RuntimeError("fedicra_amd.ops.stream_beside: torch's stream pool returned the current stream 64 times")


def _defer_wgrad_reduce(ws, stride, slices, dw, n_dw, db, cout, keep=None, cin3=0):
    """cin3 > 0: the slices are a one-launch 3x3x3 gradient [cout][9][3][cin3] and dw the parameter's own [cout][cin3][3][3][3]."""
    global _wgrad_cb_queued
    _pending_wgrad.append((ws, stride, slices, dw, n_dw, db, cout, keep, cin3))
    if not _wgrad_cb_queued:
        # runs once, after the last node of the current backward pass
        torch.autograd.Variable._execution_engine.queue_callback(flush_wgrad)
        _wgrad_cb_queued = True


def flush_wgrad():
    """Second stage of the deterministic wgrad for every layer of this backward pass: ONE launch that adds the
    per-workgroup partial slices, in fixed order, into the (flat) weight / bias gradients."""
    global _wgrad_cb_queued
    _wgrad_cb_queued = False
    if not _pending_wgrad:
        return
    rows, nblocks, nblocks3d = [], 0, 0
    for ws, stride, slices, dw, n_dw, db, cout, _keep, cin3 in _pending_wgrad:
        ll = 8 if slices <= 16 else 6 if slices <= 64 else 4      # fewer lanes per row when there are many slices to fold
        # 3x3x3 layers (cin3 > 0): word 9 < 0 -- the reduce leaves their sums in slice 0 and fi_wgrad_permute3d_multi, one
        # workgroup per (output channel, 64 input channels), adds them into the parameter layout (include/fedicra_hip.h)
        rows.append([ws.data_ptr(), stride, slices, dw.data_ptr(), n_dw, 0 if db is None else db.data_ptr(), cout,
                     nblocks, ll, -cin3, nblocks3d])
        nblocks += -(-stride // (4 << ll))
        if cin3 > 0:
            nblocks3d += cout * -(-cin3 // 64)
    dev = _pending_wgrad[0][0].device
    capturing = torch.cuda.is_current_stream_capturing()
    host = _pinned_slot(len(rows), capturing)
    host.copy_(torch.tensor(rows, dtype=torch.int64))
    table = host.to(dev, non_blocking=True)
    L.wgrad_reduce_multi(table, len(rows), nblocks, nblocks3d)
    if capturing:
        # the graph re-reads the host table and the partials on every replay: they live as long as the owning client
        _ctx.graph_tables.append((host, table, [p[:7] for p in _pending_wgrad]))
    _pending_wgrad.clear()


_PIN_SLOTS, _PIN_ROWS = 32, 128
_pin_pool = None                       # eager launches: a ring of pinned staging tables
_pin_next = 0
_graph_free = []                       # captured graphs: each takes a pinned table for good (the graph re-reads it on replay)


def reserve_graph_tables(n=4):
    """Make sure `n` pinned reduce tables are available to a stream capture that is about to start.  Pinned memory
    cannot be allocated while a stream is capturing, and a captured graph keeps its table for as long as it lives, so
    the pool grows here -- outside the capture -- instead of being a fixed number of slots (a long-running process
    that builds new clients every round would run out).  Called by the trainers right before torch.cuda.graph()."""
    while len(_graph_free) < n:
        chunk = torch.empty((8, _PIN_ROWS, L.WGRAD_ROW), dtype=torch.int64).pin_memory()
        _graph_free.extend(chunk[i] for i in range(chunk.shape[0]))


def _pinned_slot(nrows, capturing):
    """Pinned host staging for the reduce table (no host allocation inside a stream capture)."""
    global _pin_pool, _pin_next
    if nrows > _PIN_ROWS:
        raise L.FiError("too many deferred wgrad reductions in one backward pass")
    if capturing:
        if not _graph_free:
            raise L.FiError("no pinned reduce table left for this capture: call ops.reserve_graph_tables() before "
                            "torch.cuda.graph()")
        return _graph_free.pop()[:nrows]
    if _pin_pool is None:
        _pin_pool = torch.empty((_PIN_SLOTS, _PIN_ROWS, L.WGRAD_ROW), dtype=torch.int64).pin_memory()
    s = _pin_next
    _pin_next = (s + 1) % _PIN_SLOTS
    return _pin_pool[s, :nrows]


def _conv_backward(ctx, dy, x0, x1, wk, mod):
    """Shared by _Conv and _ConvBNAct: returns (dx0, dx1, gw, gb).

    The weight gradient only feeds the optimizer: its first stage (per-workgroup partial sums) is issued here, in line with
    the dgrad -> BN-backward chain (beside it on a second stream measured slower, 2.59 against 2.40 ms per step: round 2), the
    single multi-tensor second stage at the end of backward."""
    ksize, cout, cin = ctx.ksize, wk.shape[0], wk.shape[3]
    kk = ksize * ksize
    dx0 = dx1 = gw = gb = None
    need_x0, need_x1 = ctx.need_x0, ctx.need_x1
    if ctx.need_w or ctx.need_b:
        if ctx.need_w:
            wt_, fresh, gw = _grad_target(mod.weight)
            dw = wt_.permute(0, 2, 3, 1)
            if not dw.is_contiguous():            # foreign layout: accumulate in a KRSC scratch
                scratch = torch.zeros(wk.shape, dtype=torch.float32, device=dy.device)
                dw = scratch
        else:
            dw = torch.zeros(wk.shape, dtype=torch.float32, device=dy.device)   # kernel needs a dw target
        db = None
        if ctx.need_b:
            db, _, gb = _grad_target(mod.bias)
        if ctx.need_w and dw.data_ptr() == wt_.data_ptr() and gw is None and gb is None:
            # stage 1 now (partial sums into a workspace); stage 2 of ALL layers in one launch at the end of backward.
            # Only when both targets are gradient SINKS this function does not hand back to autograd (a flat-store model,
            # or .grad already in place): a fresh tensor returned from backward may be cloned by AccumulateGrad before the
            # deferred reduce has written it -- a plain nn.Conv2d with a 1x1 kernel (whose KRSC view is contiguous) used to
            # get zeros for dw and db that way (tools/grad_bisect.py, round 5)
            ws, slices, stride = L.conv2d_wgrad_partial(x0, x1, dy, db is not None, ksize=ksize)
            _defer_wgrad_reduce(ws, stride, slices, dw, cout * kk * cin, db, cout, None)
        else:
            L.conv2d_wgrad(x0, x1, dy, dw, db, ksize=ksize)
            if ctx.need_w and dw.data_ptr() != wt_.data_ptr():
                wt_.add_(dw.permute(0, 3, 1, 2))
    if need_x0 or need_x1:
        wt = _packed(wk, dy.dtype, 1, cout, kk, cin, param=mod.weight)
        N, H, W, _ = dy.shape
        c0 = x0.shape[3]
        if x1 is not None and need_x0 != need_x1:
            # only ONE half of a concatenated input wants its gradient (FedICRA's ALA: the skip connection comes from the
            # frozen encoder, flower_common.py:542-546): the dgrad operand's rows are input channels, so that half is a
            # contiguous row range of it -- half the GEMM instead of computing a tensor nobody reads
            rows = slice(0, c0) if need_x0 else slice(c0, cin)
            dh = torch.empty((N, H, W, rows.stop - rows.start), dtype=dy.dtype, device=dy.device)
            L.conv2d_fwd(dy, None, wt.view(cin, kk * cout)[rows].reshape(-1), None, dh, None, None, ksize=ksize, tag="conv_dgrad")
            dx0, dx1 = (dh, None) if need_x0 else (None, dh)
        else:
            # dgrad = conv of dy with the flipped/transposed filter; two destinations for a concat input
            d0 = torch.empty((N, H, W, c0), dtype=dy.dtype, device=dy.device)
            d1 = None if x1 is None else torch.empty((N, H, W, x1.shape[3]), dtype=dy.dtype, device=dy.device)
            L.conv2d_fwd(dy, None, wt, None, d0, d1, None, ksize=ksize, tag="conv_dgrad")
            dx0 = d0 if need_x0 else None
            dx1 = d1 if need_x1 else None
    return dx0, dx1, gw, gb


class _Conv(Function):
    """y = conv(cat(x0,x1)) + bias  (out_conv, conv1x1, head 1x1, PCS fc layers)."""

    @staticmethod
    def forward(ctx, x0, x1, weight, bias, mod, y_f32):
        wk = _krsc(weight)
        cout, ksize, cin = wk.shape[0], wk.shape[1], wk.shape[3]
        N, H, W, _ = x0.shape
        wp = _packed(wk, x0.dtype, 0, cout, ksize * ksize, cin, param=weight)
        y = torch.empty((N, H, W, cout), dtype=torch.float32 if y_f32 else x0.dtype, device=x0.device)
        L.conv2d_fwd(x0, x1, wp, bias, y, None, None, ksize=ksize, y_f32=y_f32)
        ctx.save_for_backward(x0, x1, wk)
        ctx.mod, ctx.ksize, ctx.y_f32 = mod, ksize, y_f32
        ctx.need_x0 = ctx.needs_input_grad[0]
        ctx.need_x1 = x1 is not None and ctx.needs_input_grad[1]
        ctx.need_w = ctx.needs_input_grad[2]
        ctx.need_b = bias is not None and ctx.needs_input_grad[3]
        return y

    @staticmethod
    def backward(ctx, dy):
        x0, x1, wk = ctx.saved_tensors
        dy = dy.contiguous()
        if dy.dtype != x0.dtype:
            t = torch.empty(dy.shape, dtype=x0.dtype, device=dy.device)
            L.cast(dy, t)
            dy = t
        dx0, dx1, gw, gb = _conv_backward(ctx, dy, x0, x1, wk, ctx.mod)
        return dx0, dx1, gw, gb, None, None


class _ConvBNAct(Function):
    """z = dropout(act(BN(conv(cat(x0,x1)) + bias)))  -- one ConvBlock half (unet.py:19-27)."""

    @staticmethod
    def forward(ctx, x0, x1, weight, bias, gamma, beta, mod, bn, slope, drop_p, drop_kind):
        wk = _krsc(weight)
        cout, ksize, cin = wk.shape[0], wk.shape[1], wk.shape[3]
        N, H, W, _ = x0.shape
        dev = x0.device
        training = bn.training
        wp = _packed(wk, x0.dtype, 0, cout, ksize * ksize, cin, param=weight)
        y = torch.empty((N, H, W, cout), dtype=x0.dtype, device=dev)
        stats = _ctx.arena.take(L.STATS_SLOTS * cout * 2, dev) if training else None
        L.conv2d_fwd(x0, x1, wp, bias, y, None, stats, ksize=ksize)
        coef = torch.empty(4, cout, dtype=torch.float32, device=dev)      # scale, shift, mean, invstd
        drop = _drop_spec(drop_p, drop_kind, N, H, W, cout, dev, owner=bn) if training else None
        z = torch.empty_like(y)
        # BN finalize (batch statistics -> scale/shift, running-stat update) + apply + activation + dropout: 1 launch
        L.bn_fused_fwd(y, z, stats, gamma, beta, bn.running_mean, bn.running_var, bn.num_batches_tracked, bn.momentum,
                       bn.eps, training, coef, slope, drop)
        ctx.save_for_backward(x0, x1, wk, y, coef)
        ctx.mod, ctx.bn, ctx.ksize, ctx.slope, ctx.drop, ctx.training = mod, bn, ksize, slope, drop, training
        ctx.need_x0 = ctx.needs_input_grad[0]
        ctx.need_x1 = x1 is not None and ctx.needs_input_grad[1]
        ctx.need_w = ctx.needs_input_grad[2]
        ctx.need_b = bias is not None and ctx.needs_input_grad[3]
        ctx.need_g, ctx.need_be = ctx.needs_input_grad[4], ctx.needs_input_grad[5]
        return z

    @staticmethod
    def backward(ctx, dz):
        x0, x1, wk, y, coef = ctx.saved_tensors
        dz = dz.contiguous()
        cout = y.shape[3]
        sums = _ctx.arena.take(L.STATS_SLOTS * cout * 2, y.device)
        L.bn_act_bwd_reduce(dz, y, coef[0], coef[1], coef[2], coef[3], sums, ctx.slope, ctx.drop)
        gg = gbeta = None
        dgam = dbet = None
        if ctx.need_g:
            dgam, _, gg = _grad_target(ctx.bn.weight)       # fresh targets are zeroed -> always accumulate
        if ctx.need_be:
            dbet, _, gbeta = _grad_target(ctx.bn.bias)
        need_dy = ctx.need_x0 or ctx.need_x1 or ctx.need_w or ctx.need_b
        dy = torch.empty_like(y) if need_dy else None
        L.bn_act_bwd_apply(dz, y, coef[0], coef[1], coef[2], coef[3], sums, ctx.training, dy, dgam, dbet, ctx.slope,
                           ctx.drop, accumulate_param=True)
        dx0 = dx1 = gw = gb = None
        if need_dy:
            dx0, dx1, gw, gb = _conv_backward(ctx, dy, x0, x1, wk, ctx.mod)
        return dx0, dx1, gw, gb, gg, gbeta, None, None, None, None, None


class _MaxPool(Function):
    @staticmethod
    def forward(ctx, x):
        N, H, W, Cc = x.shape
        y = torch.empty((N, H // 2, W // 2, Cc), dtype=x.dtype, device=x.device)
        L.maxpool2_fwd(x, y)
        ctx.save_for_backward(x)
        return y

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        dx = torch.empty_like(x)
        L.maxpool2_bwd(x, dy.contiguous(), dx)
        return dx


class _PoolSkip(Function):
    """(skip, pooled) = (z, maxpool2(z)) for an encoder feature that feeds both the next level and a decoder skip
    connection (unet.py:91-99: x0..x3 are returned AND pooled).  autograd would add the two gradients with a separate
    elementwise kernel; backward here writes skip-gradient + scattered pooling gradient in the max-pool pass itself
    (same rounding: one fp32 add per element, then the storage dtype)."""

    @staticmethod
    def forward(ctx, x):
        N, H, W, Cc = x.shape
        y = torch.empty((N, H // 2, W // 2, Cc), dtype=x.dtype, device=x.device)
        L.maxpool2_fwd(x, y)
        ctx.save_for_backward(x)
        return x.view_as(x), y

    @staticmethod
    def backward(ctx, dskip, dy):
        (x,) = ctx.saved_tensors
        if dy is None:
            return dskip
        dx = torch.empty_like(x)
        if dskip is None:
            L.maxpool2_bwd(x, dy.contiguous(), dx)
        else:
            if dskip.dtype != x.dtype:
                dskip = dskip.to(x.dtype)
            L.maxpool2_bwd_add(x, dy.contiguous(), dskip.contiguous(), dx)
        return dx


class _Upsample(Function):
    @staticmethod
    def forward(ctx, x):
        N, h, w, Cc = x.shape
        y = torch.empty((N, 2 * h, 2 * w, Cc), dtype=x.dtype, device=x.device)
        L.upsample2x_fwd(x, y)
        ctx.shape = x.shape
        return y

    @staticmethod
    def backward(ctx, dy):
        dx = torch.empty(ctx.shape, dtype=dy.dtype, device=dy.device)
        L.upsample2x_bwd(dy.contiguous(), dx)
        return dx


_UPFUSE = os.environ.get("FI_UPFUSE", "1") != "0"              # measurement switch: 0 = conv1x1 and up-sampling as two launches


class _ConvUp(Function):
    """u = Upsample2x(conv1x1(x) + bias): UpBlock's first half (unet.py:57-59,65-67) as ONE forward launch
    (fi_conv1x1_up2x_fwd: the low-resolution convolution output lives in LDS only).  Backward is the two ops' own:
    fi_upsample2x_bwd, then the 1x1 convolution's input / filter gradients."""

    @staticmethod
    def forward(ctx, x0, weight, bias, mod):
        wk = _krsc(weight)
        cout, cin = wk.shape[0], wk.shape[3]
        N, h, w, _ = x0.shape
        wp = _packed(wk, x0.dtype, 0, cout, 1, cin, param=weight)
        u = torch.empty((N, 2 * h, 2 * w, cout), dtype=x0.dtype, device=x0.device)
        if not (_UPFUSE and x0.dtype != torch.float32 and L.conv1x1_up2x_fwd(x0, None, wp, bias, u)):
            y = torch.empty((N, h, w, cout), dtype=x0.dtype, device=x0.device)
            L.conv2d_fwd(x0, None, wp, bias, y, None, None, ksize=1)
            L.upsample2x_fwd(y, u)
        ctx.save_for_backward(x0, wk)
        ctx.mod, ctx.ksize = mod, 1
        ctx.need_x0, ctx.need_x1 = ctx.needs_input_grad[0], False
        ctx.need_w = ctx.needs_input_grad[1]
        ctx.need_b = bias is not None and ctx.needs_input_grad[2]
        return u

    @staticmethod
    def backward(ctx, du):
        x0, wk = ctx.saved_tensors
        N, h, w, _ = x0.shape
        dy = torch.empty((N, h, w, wk.shape[0]), dtype=du.dtype, device=du.device)
        L.upsample2x_bwd(du.contiguous(), dy)
        if dy.dtype != x0.dtype:
            t = torch.empty(dy.shape, dtype=x0.dtype, device=dy.device)
            L.cast(dy, t)
            dy = t
        dx0, _, gw, gb = _conv_backward(ctx, dy, x0, None, wk, ctx.mod)
        return dx0, gw, gb, None


def conv1x1_up(x, mod):
    """UpBlock.up(UpBlock.conv1x1(x)) (bilinear decoder)."""
    return _ConvUp.apply(x, mod.weight, mod.bias, mod)


def probe_conv_up(s0, conv, groups):
    """The same in the batched no-grad forward, of a possibly raw source."""
    r0 = s0 if isinstance(s0, RawAct) else None
    x = r0.y if r0 is not None else s0
    wk = _krsc(conv.weight)
    cout, cin = wk.shape[0], wk.shape[3]
    N, h, w, _ = x.shape
    if _UPFUSE and x.dtype != torch.float32 and not (r0 is not None and r0.shared):
        u = torch.empty((N, 2 * h, 2 * w, cout), dtype=x.dtype, device=x.device)
        t0 = None if r0 is None else L.in_xform(r0.coef, r0.slope)
        if L.conv1x1_up2x_fwd(x, t0, _packed(wk, x.dtype, 0, cout, 1, cin, param=conv.weight), conv.bias, u, groups=groups):
            return u
    return upsample2x(probe_conv(s0, conv, groups))


class _CELoss(Function):
    """CrossEntropyLoss(ignore_index) on fp32 NHWC logits [N,H,W,C]; labels uint8 [N,H,W]."""

    @staticmethod
    def forward(ctx, logits, labels, ignore_index, grad_dtype):
        acc = torch.zeros(2 * L.CE_SLOTS, dtype=torch.float64, device=logits.device)
        L.ce_fwd(logits, labels, ignore_index, acc)
        loss = torch.empty(1, dtype=torch.float32, device=logits.device)
        L.ce_finalize(acc, loss)
        ctx.save_for_backward(logits, labels, acc)
        ctx.ignore, ctx.grad_dtype = ignore_index, grad_dtype
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        logits, labels, acc = ctx.saved_tensors
        dl = torch.empty(logits.shape, dtype=ctx.grad_dtype, device=logits.device)
        gs = g.reshape(1).to(torch.float32).contiguous()
        L.ce_bwd(logits, labels, ctx.ignore, acc, gs, dl)
        return dl, None, None, None


class _ChannelGate(Function):
    """PersonalizedChannelSelection tail (unet.py:142): y = x*h + x, h [N,C] fp32."""

    @staticmethod
    def forward(ctx, x, h, amax, davg_src):
        y = torch.empty_like(x)
        L.channel_gate_fwd(x, h, y)
        ctx.save_for_backward(x, h)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, h = ctx.saved_tensors
        dx = torch.empty_like(x)
        dh = torch.empty_like(h)
        L.channel_gate_bwd(x, dy.contiguous(), h, None, None, None, dx, dh)
        return dx, dh, None, None


class _GlobalAvgMax(Function):
    """AdaptiveAvgPool2d(1) and AdaptiveMaxPool2d(1) in one pass (unet.py:125-126) -> ([N,C], [N,C]) fp32."""

    @staticmethod
    def forward(ctx, x):
        N, H, W, Cc = x.shape
        avg = torch.empty((N, Cc), dtype=torch.float32, device=x.device)
        mx = torch.empty_like(avg)
        amax = torch.empty((N, Cc), dtype=torch.int32, device=x.device)
        L.global_avgmax(x, avg, mx, amax)
        ctx.save_for_backward(amax)
        ctx.xshape, ctx.xdtype = x.shape, x.dtype
        return avg, mx

    @staticmethod
    def backward(ctx, davg, dmx):
        (amax,) = ctx.saved_tensors
        N, H, W, Cc = ctx.xshape
        # dx = davg/HW everywhere + dmx at the arg-max pixel: channel_gate_bwd with dy = 0, h = -1
        zero = torch.zeros(ctx.xshape, dtype=ctx.xdtype, device=davg.device)
        hneg = torch.full((N, Cc), -1.0, dtype=torch.float32, device=davg.device)
        dx = torch.empty_like(zero)
        L.channel_gate_bwd(zero, zero, hneg, amax, davg.contiguous(), dmx.contiguous(), dx, None)
        return dx


class _PCSGate(Function):
    """h = sigmoid(fc2([avg ; e]) + fc2([max ; e])), e = fc1(onehot(who))  (unet.py:122-141) -- fi_pcs_gate_fwd / _bwd.  The PCS
    weights are frozen in the reference (unregistered, never optimised): backward yields d/d avg and d/d max only."""

    @staticmethod
    def forward(ctx, avg, mx, who, w1a, w1b, w2a, w2b):
        B, Cc = avg.shape
        h = torch.empty_like(avg)
        hidden = torch.empty((B, 2, Cc // 16), dtype=torch.float32, device=avg.device)
        L.pcs_gate_fwd(avg.contiguous(), mx.contiguous(), who, w1a, w1b, w2a, w2b, h, hidden)
        ctx.save_for_backward(h, hidden, w2a, w2b)
        return h

    @staticmethod
    def backward(ctx, dh):
        h, hidden, w2a, w2b = ctx.saved_tensors
        davg, dmx = torch.empty_like(h), torch.empty_like(h)
        L.pcs_gate_bwd(dh.contiguous(), h, hidden, w2a, w2b, davg, dmx)
        return davg, dmx, None, None, None, None, None


def pcs_gate(avg, mx, who, fc1, fc2):
    """avg / mx fp32 [B,C]; who int32 [B]; fc1 / fc2: the nn.Sequential(Conv2d 1x1, ReLU, Conv2d 1x1) pairs of the module."""
    f = lambda m: m.weight.detach().reshape(m.weight.shape[0], m.weight.shape[1]).float().contiguous()
    return _PCSGate.apply(avg, mx, who, f(fc1[0]), f(fc1[2]), f(fc2[0]), f(fc2[2]))


class _LCTotal(Function):
    """(total, loss_lc) = (loss_ce + alpha * loss_lc, -(1/G) sum_g mse(h, o_g))  (flower_pCE_2D.py:128-139), one launch each way."""

    @staticmethod
    def forward(ctx, loss_ce, h, others, G, alpha):
        out = torch.empty(2, dtype=torch.float32, device=h.device)
        dcoef = torch.empty(h.numel(), dtype=torch.float32, device=h.device)
        L.lc_loss_fwd(h.contiguous(), others, loss_ce.reshape(1), alpha, G, out, dcoef)
        ctx.save_for_backward(dcoef)
        ctx.alpha, ctx.hshape = alpha, h.shape
        total, lc = out[0], out[1]
        ctx.mark_non_differentiable(lc)
        return total, lc

    @staticmethod
    def backward(ctx, g, _g_lc):
        (dcoef,) = ctx.saved_tensors
        dh = torch.empty(ctx.hshape, dtype=torch.float32, device=dcoef.device)
        gs = g.reshape(1).to(torch.float32).contiguous()
        L.lc_loss_bwd(dcoef, gs, ctx.alpha, dh)
        return g, dh, None, None, None


def lc_total(loss_ce, h, others, G, alpha):
    """h: the client's own heat-map (any shape, fp32, n elements); others: ONE fp32 tensor holding the G other heat-maps back to
    back ([G * n] elements, no gradient).  -> (loss_ce + alpha * loss_lc, loss_lc)."""
    return _LCTotal.apply(loss_ce, h.float(), others, int(G), float(alpha))


# ----------------------------------------------------------------------------- functional API
def conv2d(x0, x1, mod, y_f32=False):
    return _Conv.apply(x0, x1, mod.weight, mod.bias, mod, y_f32)


def conv_bn_stats_only(x0, x1, conv, bn, drop_p=0.0, drop_kind="elem"):
    """The side effects of a train-mode ``BN(conv(x))`` whose OUTPUT nobody reads (no-grad forwards of FedICRA's LC loss
    through the auxiliary heads, flower_pCE_2D.py:128-139): the convolution runs with its statistics epilogue but stores
    nothing, and the finalize kernel moves running_mean / running_var / num_batches_tracked exactly as the fused
    forward would.  Only meaningful without autograd."""
    assert bn.training and not torch.is_grad_enabled()
    if _mask_provider is not None and drop_p > 0.0:
        # parity mode draws masks from the host generator in forward order: the skipped dropout still takes its draw,
        # or every later mask of the run would shift against the reference's sequence
        N_, H_, W_, _ = x0.shape
        _mask_provider((N_, conv.weight.shape[0], H_, W_) if drop_kind == "elem" else (N_, conv.weight.shape[0], 1, 1), drop_p)
    wk = _krsc(conv.weight)
    cout, ksize, cin = wk.shape[0], wk.shape[1], wk.shape[3]
    N, H, W, _ = x0.shape
    dev = x0.device
    wp = _packed(wk, x0.dtype, 0, cout, ksize * ksize, cin, param=conv.weight)
    stats = _ctx.arena.take(L.STATS_SLOTS * cout * 2, dev)
    L.conv2d_fwd(x0, x1, wp, conv.bias, None, None, stats, ksize=ksize, cout=cout)
    coef = torch.empty(4, cout, dtype=torch.float32, device=dev)
    L.bn_finalize(stats, float(N * H * W), bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.num_batches_tracked,
                  bn.momentum, bn.eps, True, coef[0], coef[1], coef[2], coef[3])


# ----------------------------------------------------------------------------- fused probe forward
_XCORR = os.environ.get("FI_XCORR", "1") != "0"                # measurement switch: 0 = the direct statistics-only launch
_POOL_MAT = os.environ.get("FI_POOL_MAT", "1") != "0"          # measurement switch: 0 = pooled sources stay in the loader
_POOL_MAT_MIN_CIN = int(os.environ.get("FI_POOL_MAT_MIN_CIN", "64"))
_XCORR_MIN_COUT = int(os.environ.get("FI_XCORR_MIN_COUT", "192"))   # 13 x 64 x 64 vs 9 x 64 x Cout multiply-adds: pays from Cout ~ 2 x 92 on


class RawAct:
    """A ConvBlock half whose activation z = dropout(act(BN(y))) is never materialised: the raw convolution output y, the
    coefficient rows of its BatchNorm (fp32 [2][G][C], one row pair per statistics group) and the activation slope.  The
    consuming convolution applies them in its loader (fi_conv2d_fwd_fused).  `shared`: y holds ONE group's images, which
    every group reads."""
    __slots__ = ("y", "coef", "slope", "shared")

    def __init__(self, y, coef, slope, shared=False):
        self.y, self.coef, self.slope, self.shared = y, coef, slope, shared


def run_probe_deferred(items):
    """The running-statistics updates the batched forward left to the caller (_probe_finalize), on the CURRENT stream: all
    layers in one launch (fi_bn_running_groups_multi)."""
    if not items:
        return
    cur = torch.cuda.current_stream()
    for it in items:                                   # the statistics were produced on the probe stream and are read here, on another one:
        it[0].record_stream(cur)                       # tell the caching allocator (a buffer that is not the arena's could be handed out again)
    with torch.no_grad():
        L.bn_running_groups_multi(items)
    items.clear()


def probe_ready():
    """The batched probe forward draws its masks from the device RNG stream of the current context (a training client's
    iteration counter); parity runs that feed host masks take the sequential path instead."""
    return _mask_provider is None and _ctx.seed_offset is not None


def _probe_drop(p, bn_owner, groups):
    """Dropout of a ConvBlock's first half for `groups` consecutive calls of the layer: the spec of the FIRST call (ops._drop_spec
    numbering: seed = base + uid<<24 + k*0x10001) -- group g draws with k + g -- and the call counter moved past all of them."""
    if p <= 0.0:
        return None
    spec = _drop_spec(p, "elem", 1, 1, 1, 1, None, owner=bn_owner)
    uid = bn_owner._fi_uid if getattr(bn_owner, "_fi_uid", None) is not None else _layer_uid[id(bn_owner)]
    _ctx.call_idx[uid] += groups - 1
    return spec


def probe_conv_bn(s0, s1, conv, bn, slope, groups, *, pool=False, in_drop=None, store=True):
    """One ConvBlock half of the batched no-grad train-mode forward: convolution of (possibly raw) sources with the
    statistics epilogue per group, then the grouped BatchNorm finalize (running statistics moved `groups` times, in
    order).  Returns RawAct(y, coef) -- or None when store is False (statistics-only: the auxiliary heads)."""
    r0 = s0 if isinstance(s0, RawAct) else None
    r1 = s1 if isinstance(s1, RawAct) else None
    x0 = r0.y if r0 is not None else s0
    x1 = None if s1 is None else (r1.y if r1 is not None else s1)
    wk = _krsc(conv.weight)
    cout, ksize, cin = wk.shape[0], wk.shape[1], wk.shape[3]
    dev = x0.device
    shared0 = bool(r0 is not None and r0.shared)
    N = x0.shape[0] * (groups if shared0 else 1)
    H, W = x0.shape[1], x0.shape[2]
    if pool:
        H, W = H // 2, W // 2
    wp = _packed(wk, x0.dtype, 0, cout, ksize * ksize, cin, param=conv.weight)
    if (pool and _POOL_MAT and r0 is not None and not shared0 and in_drop is None and s1 is None and x0.dtype != torch.float32
            and cin >= _POOL_MAT_MIN_CIN):
        # deep DownBlock (64+ channels in): the pooled activation written out once (fi_bn_act_pool_groups: the loader's own
        # arithmetic), then the plain loader of the wave-specialised kernels instead of transforming and pooling the tile once per
        # output slab in the one-tile kernel -- 84 x 32^2 128 -> 256: 188 us at 0.11 of its roofline before
        z = torch.empty((x0.shape[0], H, W, cin), dtype=x0.dtype, device=dev)
        L.bn_act_pool_groups(x0, r0.coef, r0.slope, z, groups)
        x0, r0, pool = z, None, False
    t0 = None if r0 is None else L.in_xform(r0.coef, r0.slope, pool=pool, drop=in_drop, seed_group_stride=0x10001)
    t1 = None if r1 is None else L.in_xform(r1.coef, r1.slope)
    if (pool or in_drop is not None) and t0 is None:
        raise L.FiError("pooling / dropout in the loader need a raw source 0")
    stats = _ctx.arena.take(groups * L.STATS_SLOTS * cout * 2, dev)
    y = torch.empty((N, H, W, cout), dtype=x0.dtype, device=dev) if store else None
    done = False
    if (not store and _XCORR and ksize == 3 and x1 is None and not shared0 and not pool and in_drop is None
            and x0.dtype != torch.float32 and cin == 64 and cout >= _XCORR_MIN_COUT):
        # statistics-only launch of a wide layer on a 64-channel input (the auxiliary head of the LC forwards, 64 -> 512): the
        # statistics from the input's autocorrelation instead of the convolution nobody reads (csrc/xcorr.hip).  fp32 parity
        # mode, other shapes and FI_XCORR=0 keep the direct launch; False = shape not covered, nothing was launched
        done = L.conv2d_stats_xcorr(x0, t0, wp, conv.bias, stats, groups=groups, cout=cout)
    if not done:
        L.conv2d_fwd_fused(x0, t0, x1, t1, wp, conv.bias, y, stats, ksize=ksize, groups=groups, cout=cout, shared0=shared0)
    count = float((N // groups) * H * W)
    coef = torch.empty((2, groups, cout), dtype=torch.float32, device=dev) if store else None
    _probe_finalize(stats, groups, count, bn, coef)
    return RawAct(y, coef, slope) if store else None


def _probe_finalize(stats, groups, count, bn, coef, shared=False):
    """Grouped BatchNorm finalize of the batched no-grad forward: coefficient rows for the consuming loader and `groups`
    running-statistics updates in group order.  Beside the client's own forward (`_ctx.probe_deferred` is a list) only the
    COEFFICIENTS are made here -- they depend on the batch statistics alone -- and the running-statistics half of the same
    kernel is handed to the caller, who runs it on ITS stream after joining this one: behind the own forward's update of the
    same BatchNorm by stream order (the reference's order, flower_pCE_2D.py:106,128-139), with NO edge between the two
    branches of the captured step.  (Round 3 ordered the two updates with one event per layer; a single such edge late in the
    step measured 1 398 -> 1 282 images/s on one box in round 5, and dropping all of them is what this form is for.)"""
    if _ctx.probe_deferred is not None:
        if coef is not None:
            L.bn_finalize_groups(stats, groups, count, bn.weight, bn.bias, None, None, None, bn.momentum, bn.eps, coef, shared=shared)

        # (stats, groups, count, running_mean, running_var, num_batches_tracked, momentum, shared): run_probe_deferred()
        _ctx.probe_deferred.append((stats, groups, count, bn.running_mean, bn.running_var, bn.num_batches_tracked,
                                    bn.momentum, shared))
        return
    if coef is None:
        coef = torch.empty((2, groups, bn.weight.numel()), dtype=torch.float32, device=stats.device)
    L.bn_finalize_groups(stats, groups, count, bn.weight, bn.bias, bn.running_mean, bn.running_var,
                         bn.num_batches_tracked, bn.momentum, bn.eps, coef, shared=shared)


def probe_first_conv_bn(x, conv, bn, slope, groups):
    """The first convolution of the batch: the same images under every embedding, so it runs ONCE; its batch statistics
    are every group's, and the running statistics still move `groups` times."""
    wk = _krsc(conv.weight)
    cout, ksize, cin = wk.shape[0], wk.shape[1], wk.shape[3]
    N, H, W, _ = x.shape
    dev = x.device
    wp = _packed(wk, x.dtype, 0, cout, ksize * ksize, cin, param=conv.weight)
    stats = _ctx.arena.take(L.STATS_SLOTS * cout * 2, dev)
    y = torch.empty((N, H, W, cout), dtype=x.dtype, device=dev)
    L.conv2d_fwd(x, None, wp, conv.bias, y, None, stats, ksize=ksize)
    coef = torch.empty((2, groups, cout), dtype=torch.float32, device=dev)
    _probe_finalize(stats, groups, float(N * H * W), bn, coef, shared=True)
    return RawAct(y, coef, slope, shared=True)


def probe_conv(s0, conv, groups):
    """Plain convolution (UpBlock.conv1x1) of a possibly raw source."""
    if not isinstance(s0, RawAct):
        wk = _krsc(conv.weight)
        cout, ksize, cin = wk.shape[0], wk.shape[1], wk.shape[3]
        N, H, W, _ = s0.shape
        y = torch.empty((N, H, W, cout), dtype=s0.dtype, device=s0.device)
        L.conv2d_fwd(s0, None, _packed(wk, s0.dtype, 0, cout, ksize * ksize, cin, param=conv.weight), conv.bias, y, None,
                     None, ksize=ksize)
        return y
    wk = _krsc(conv.weight)
    cout, ksize, cin = wk.shape[0], wk.shape[1], wk.shape[3]
    N, H, W, _ = s0.y.shape
    y = torch.empty((N, H, W, cout), dtype=s0.y.dtype, device=s0.y.device)
    L.conv2d_fwd_fused(s0.y, L.in_xform(s0.coef, s0.slope), None, None,
                       _packed(wk, s0.y.dtype, 0, cout, ksize * ksize, cin, param=conv.weight), conv.bias, y, None,
                       ksize=ksize, groups=groups, cout=cout)
    return y


def probe_materialize(r, groups):
    """z = act(BN(y)) of a raw activation as a tensor (the deepest encoder level, whose consumer is the channel
    selection; every level for the ALA epoch's frozen encoder): all statistics groups in one launch."""
    z = torch.empty_like(r.y)
    L.bn_act_pool_groups(r.y, r.coef, r.slope, z, groups, pool=False)
    return z


def conv_bn_act(x0, x1, conv, bn, slope, drop_p=0.0, drop_kind="elem"):
    return _ConvBNAct.apply(x0, x1, conv.weight, conv.bias, bn.weight, bn.bias, conv, bn, float(slope),
                            float(drop_p), drop_kind)


def maxpool2(x):
    return _MaxPool.apply(x)


def pool_skip(x):
    """Returns (x, maxpool2(x)); use the first result wherever x is consumed besides the pooling."""
    return _PoolSkip.apply(x)


def upsample2x(x):
    return _Upsample.apply(x)


def ce_loss(logits_nhwc, labels, ignore_index):
    """logits: fp32 dense NHWC [N,H,W,C]; labels [N,H,W] (any integer dtype, values <= 255)."""
    grad_dtype = torch.float32
    if labels.dtype != torch.uint8:
        labels = labels.to(torch.uint8)
    return _CELoss.apply(logits_nhwc, labels.contiguous(), int(ignore_index), grad_dtype)


def global_avgmax(x):
    return _GlobalAvgMax.apply(x)


def channel_gate(x, h):
    return _ChannelGate.apply(x, h, None, None)


def to_nhwc(x_nchw, dtype):
    """Reference-layout fp32 [N,C,H,W] (any strides) -> dense NHWC in `dtype` (no grad)."""
    x = x_nchw.detach()
    if x.dtype != torch.float32:
        x = x.float()
    x = x.contiguous()
    N, Cc, H, W = x.shape
    out = torch.empty((N, H, W, Cc), dtype=dtype, device=x.device)
    L.nchw_to_nhwc(x, out)
    return out
