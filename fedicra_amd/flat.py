"""Flat, HBM-resident model state (parameters + BatchNorm statistics + gradients).

The reference round-trips ~140 separate arrays through numpy twice per round
(/root/reference/code/flower_common.py:488-489, 499-502).  Here every registered parameter and
float buffer of a model is a *view* into ONE fp32 device buffer (``flat_state``), gradients are
views into a second one (``flat_grads``) and the int64 ``num_batches_tracked`` counters into a
third: the fused AdamW, the FedAvg pre-scale and the RCCL all-reduce each touch one contiguous
buffer, while ``state_dict()`` / ``get_weights`` still see the reference's named tensors.
Conv weights keep their logical [Cout,Cin,k,k] shape with channels_last strides, i.e. the
[Cout][k][k][Cin] memory the implicit-GEMM kernels consume directly.
"""
from __future__ import annotations

import torch
import torch.nn as nn

_ALIGN = 8  # elements (32 B)


def _round_up(n, a=_ALIGN):
    return (n + a - 1) // a * a


def _as_view(flat, off, like):
    """View of flat[off:off+numel] with `like`'s logical shape; 4-D tensors get channels_last strides."""
    n = like.numel()
    seg = flat[off:off + n]
    if like.dim() == 4:
        co, ci, kh, kw = like.shape
        return seg.view(co, kh, kw, ci).permute(0, 3, 1, 2)
    return seg.view(like.shape)


class FlatStoreMixin:
    """Mixin for root nn.Modules.  Call ``_fi_finish_init()`` at the end of ``__init__``."""

    def _fi_finish_init(self):
        self._fi_flat_ready = False
        for i, m in enumerate(self.modules()):     # stable per-layer ids for the dropout seed stream (ops._drop_spec)
            if isinstance(m, nn.BatchNorm2d):
                m._fi_uid = i + 1
        self._fi_reflatten()

    # -- construction ---------------------------------------------------------------------------
    def _fi_reflatten(self):
        params = list(self.named_parameters())
        fbufs = [(n, b) for n, b in self.named_buffers() if b.is_floating_point()]
        ibufs = [(n, b) for n, b in self.named_buffers() if not b.is_floating_point()]
        if not params:
            return
        dev = params[0][1].device
        offs, cur = {}, 0
        for n, t in params:
            offs[n] = cur
            cur += _round_up(t.numel())
        n_param = cur
        for n, t in fbufs:
            offs[n] = cur
            cur += _round_up(t.numel())
        flat = torch.zeros(cur, dtype=torch.float32, device=dev)
        grads = torch.zeros(n_param, dtype=torch.float32, device=dev)
        inbt = torch.zeros(max(len(ibufs), 1), dtype=torch.int64, device=dev)
        with torch.no_grad():
            for n, p in params:
                v = _as_view(flat, offs[n], p)
                v.copy_(p.detach().to(device=dev, dtype=torch.float32))
                rg = p.requires_grad
                p.data = v
                p.requires_grad_(rg)
                p.grad = None
                p._fi_gview = _as_view(grads, offs[n], p)
                p._fi_off = offs[n]
                if hasattr(p, "_fi_packs"):          # operand packs belong to the storage we just left
                    del p._fi_packs
            mods = dict(self.named_modules())
            for n, b in fbufs:
                v = _as_view(flat, offs[n], b)
                v.copy_(b.detach().to(device=dev, dtype=torch.float32))
                mname, _, bname = n.rpartition(".")
                mods[mname]._buffers[bname] = v
            for i, (n, b) in enumerate(ibufs):
                inbt[i] = b.detach().to(dev)
                mname, _, bname = n.rpartition(".")
                mods[mname]._buffers[bname] = inbt[i]
        self._fi_state = flat
        self._fi_grads = grads
        self._fi_nbt = inbt
        self._fi_nparam = n_param
        self._fi_offsets = offs
        self._fi_param_names = [n for n, _ in params]
        self.__dict__["_fi_first"] = params[0][1]      # not via setattr: must not register as a parameter
        self._fi_flat_ready = True
        self._fi_pack_key = None
        self._fi_pack_sig = None

    def _fi_check_flat(self):
        """deepcopy / foreign .data assignment break the views; detect cheaply and repair."""
        if not self._fi_flat_ready or self._fi_first.data_ptr() != self._fi_state.data_ptr():
            self._fi_reflatten()

    def _apply(self, fn, *a, **k):
        out = super()._apply(fn, *a, **k)
        if getattr(self, "_fi_flat_ready", False):
            self._fi_reflatten()
        return out

    def __deepcopy__(self, memo):
        import copy
        cls = self.__class__
        new = cls.__new__(cls)
        memo[id(self)] = new
        for k, v in self.__dict__.items():
            if k in ("_fi_state", "_fi_grads", "_fi_nbt", "_fi_first"):
                continue
            new.__dict__[k] = copy.deepcopy(v, memo)
        new._fi_flat_ready = False
        new._fi_reflatten()
        return new

    # -- conv operand packs ---------------------------------------------------------------------
    def _fi_pack_signature(self, dtype):
        from . import ops
        return (self._fi_state.data_ptr(), self._fi_state._version, ops.weights_epoch(), dtype)

    def _fi_packs_current(self, dtype):
        return getattr(self, "_fi_pack_sig", None) == self._fi_pack_signature(dtype)

    def _fi_refresh_packs(self, dtype):
        """(Re)build the forward (cast) and dgrad (flipped, transposed) operands of EVERY conv weight with one
        multi-tensor launch, iff the weights changed since the last refresh.  Called by the root forward."""
        self._fi_check_flat()
        if self._fi_packs_current(dtype):
            return
        import weakref
        from . import _lib as L
        key = (self._fi_state.data_ptr(), dtype)
        if getattr(self, "_fi_pack_key", None) != key:
            convs = [p for p in self.parameters() if p.dim() == 4]
            total = sum(p.numel() for p in convs)
            dev = self._fi_state.device
            need_fwd = dtype != torch.float32          # fp32 forward consumes the master weights directly
            buf0 = torch.empty(total if need_fwd else 0, dtype=dtype, device=dev)
            buf1 = torch.empty(total, dtype=dtype, device=dev)
            # chunk-major siblings (fi_pack_weights modes 2 / 3) of the filters the 64 x 64-wave-tile kernel can take
            # (include/fedicra_hip.h: FiConv.w16); they ride on the operand tensors as `_fi_w16`
            want2 = [need_fwd and dev.type == "cuda" and p.shape[2] == p.shape[3]
                     and L.conv_weight_chunk16(dtype, p.shape[2], p.shape[1], p.shape[0]) for p in convs]
            want3 = [dev.type == "cuda" and dtype != torch.float32 and p.shape[2] == p.shape[3]
                     and L.conv_weight_chunk16(dtype, p.shape[2], p.shape[0], p.shape[1]) for p in convs]
            buf2 = torch.empty(sum(p.numel() for p, w_ in zip(convs, want2) if w_), dtype=dtype, device=dev)
            buf3 = torch.empty(sum(p.numel() for p, w_ in zip(convs, want3) if w_), dtype=dtype, device=dev)
            rows, off, off2, off3 = [], 0, 0, 0
            ref = weakref.ref(self)
            for p, w2, w3 in zip(convs, want2, want3):
                n = p.numel()
                co, ci, kh, kw = p.shape
                v0 = buf0[off:off + n] if need_fwd else None
                v1 = buf1[off:off + n]
                v2 = v3 = None
                if w2:
                    v2 = buf2[off2:off2 + n]
                    v0._fi_w16 = v2
                    off2 += n
                if w3:
                    v3 = buf3[off3:off3 + n]
                    v1._fi_w16 = v3
                    off3 += n
                rows.append([p.data_ptr(), v0.data_ptr() if need_fwd else 0, v1.data_ptr(), co, kh * kw, ci,
                             0 if v2 is None else v2.data_ptr(), 0 if v3 is None else v3.data_ptr()])
                p._fi_packs = (ref, v0, v1)
                off += n
            self._fi_pack_bufs = (buf0, buf1, buf2, buf3)
            self._fi_pack_table = torch.tensor(rows, dtype=torch.int64).to(dev)
            self._fi_pack_n = len(rows)
            self._fi_pack_key = key
        L.pack_weights_multi(self._fi_pack_table, self._fi_pack_n, dtype)
        self._fi_pack_sig = self._fi_pack_signature(dtype)

    # -- accessors ------------------------------------------------------------------------------
    @property
    def flat_state(self):
        """fp32 [P + B]: all parameters, then all float buffers (running_mean / running_var)."""
        self._fi_check_flat()
        return self._fi_state

    @property
    def flat_params(self):
        self._fi_check_flat()
        return self._fi_state[: self._fi_nparam]

    @property
    def flat_grads(self):
        self._fi_check_flat()
        return self._fi_grads

    @property
    def flat_counters(self):
        """int64 [#BN]: num_batches_tracked of every BatchNorm, state_dict order."""
        self._fi_check_flat()
        return self._fi_nbt

    def param_ranges(self, names):
        """Contiguous [start, end) ranges of flat_params covering exactly the parameters in `names`."""
        want = set(names)
        ranges = []
        sizes = dict(self.named_parameters())
        for n in self._fi_param_names:
            if n not in want:
                continue
            s = self._fi_offsets[n]
            e = s + _round_up(sizes[n].numel())
            if ranges and ranges[-1][1] == s:
                ranges[-1][1] = e
            else:
                ranges.append([s, e])
        return [tuple(r) for r in ranges]

    def zero_grad(self, set_to_none: bool = True):
        """Same contract as nn.Module.zero_grad (grads become None); the whole flat gradient buffer is
        cleared with ONE memset and each sink is marked clean so backward() need not zero it again."""
        self._fi_check_flat()
        self._fi_grads.zero_()
        for p in self.parameters():
            p.grad = None
            p._fi_zeroed = True
