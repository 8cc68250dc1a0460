"""Fused AdamW over the model's flat parameter buffer, with every scalar on the device.

Semantics: ``torch.optim.AdamW(params, lr, betas=(0.9,0.999), eps=1e-8, weight_decay=1e-2,
amsgrad=False)`` exactly as the reference creates it at the start of every round
(/root/reference/code/flower_pCE_2D.py:55), plus the poly learning-rate schedule applied after
each iteration (:154-157).  Parameters whose ``.grad`` is None are skipped entirely (torch 2.x
``zero_grad(set_to_none=True)`` semantics -- SURVEY.md section 0 item 10), which is what makes
FedICRA's freeze schedule (:84-101) work: the first ``iters - rep_iters`` iterations only touch
``decoder.out_conv``.

``frozen="torch1"`` (``args.adamw_frozen``) selects the semantics of the reference's PINNED environment instead
(fed39v2.yaml: PyTorch 1.10.2, where ``optimizer.zero_grad()`` zeroes gradients in place): once a parameter has held a
gradient it keeps a (zero) gradient tensor for the rest of the process, and AdamW keeps stepping it while the freeze
schedule has it switched off -- decoupled weight decay, and the decay of whatever momentum the round's fresh optimizer
has already collected for it.  Every parameter that has ever received a gradient therefore steps at every iteration; a
parameter's step count starts at its first step of the round.

torch keeps one step counter per parameter; here parameters are grouped by *freeze pattern*
(the set of parameters that received a gradient), each pattern owning one device-side step
counter.  That is exact as long as the patterns seen within one round are disjoint, which holds
for FedAvg (one pattern) and FedICRA (out_conv, then everything else).

lr, the iteration counter, the per-pattern step counters and the derived AdamW scalars all live
in device memory and are advanced by one-thread kernels, so a captured hipGraph of the whole
training step can be replayed without host involvement.
"""
from __future__ import annotations

import torch

from . import _lib as L


def _adopt_foreign_grads(params, active):
    """Ops that hand autograd an ordinary gradient tensor (the 3D surface, plumbing glue) instead of accumulating into the
    parameter's sink of the flat gradient buffer: move such gradients into the sink so that the fused step sees them."""
    for n in active:
        p = params[n]
        sink = getattr(p, "_fi_gview", None)
        if sink is not None and p.grad.data_ptr() != sink.data_ptr():
            sink.copy_(p.grad.reshape(sink.shape) if p.grad.shape != sink.shape else p.grad)
            p.grad = sink


class FusedAdamW:
    MAX_GROUPS = 8

    def __init__(self, model, lr, base_lr=None, max_iterations=None, betas=(0.9, 0.999), eps=1e-8,
                 weight_decay=1e-2, bf16_shadow=False, frozen="torch2"):
        if frozen not in ("torch2", "torch1"):
            raise ValueError("FusedAdamW: frozen must be 'torch2' (grad None -> skipped) or 'torch1' (zero-grad steps)")
        self.frozen = frozen
        self._ever = set()              # torch1: parameters that hold a gradient tensor (it outlives the per-round optimizer)
        self.model = model
        self.betas, self.eps, self.wd = betas, eps, weight_decay
        self.base_lr = lr if base_lr is None else base_lr
        self.max_iterations = max_iterations
        dev = model.flat_params.device
        n = model.flat_params.numel()
        self.m = torch.zeros(n, dtype=torch.float32, device=dev)
        self.v = torch.zeros(n, dtype=torch.float32, device=dev)
        self.lr_state = torch.tensor([lr], dtype=torch.float64, device=dev)
        self.iter = torch.zeros(1, dtype=torch.int32, device=dev)
        self.steps = torch.zeros(self.MAX_GROUPS, dtype=torch.int32, device=dev)
        self.hyper = torch.zeros(self.MAX_GROUPS, 4, dtype=torch.float32, device=dev)
        self.shadow = torch.empty(n, dtype=torch.bfloat16, device=dev) if bf16_shadow else None
        self._groups = {}
        self._names = [n_ for n_, _ in model.named_parameters()]

    # -- host-visible scalars -------------------------------------------------------------------
    def set_lr(self, lr: float, current_iter: int = 0):
        self.lr_state.fill_(lr)
        self.iter.fill_(int(current_iter))

    def reset_round(self):
        """The reference builds a fresh AdamW every round: moments and step counts restart."""
        self.m.zero_()
        self.v.zero_()
        self.steps.zero_()
        self._groups = {}

    def zero_grad(self, set_to_none: bool = True):
        self.model.zero_grad(set_to_none)

    def layout_key(self):
        """The round's parameter groups so far as (counter index, #parameters, first name) -- what a captured step of the
        'torch1' semantics must agree on with the round it is replayed in (flower_pCE_2D.train_steps)."""
        return tuple((gi, len(names), names[0] if names else "") for names, (gi, _) in self._groups.items())

    # -- the step -------------------------------------------------------------------------------
    def _group_for(self, active):
        key = tuple(active)
        g = self._groups.get(key)
        if g is None:
            if len(self._groups) >= self.MAX_GROUPS:
                raise RuntimeError("FusedAdamW: too many distinct freeze patterns in one round")
            seen = set().union(*[set(k) for k in self._groups]) if self._groups else set()
            if seen & set(key):
                raise RuntimeError("FusedAdamW: overlapping freeze patterns within a round are not supported "
                                   "(per-parameter step counts would diverge from torch.optim.AdamW)")
            g = (len(self._groups), self.model.param_ranges(key))
            self._groups[key] = g
        return g

    def step(self):
        self.step_scaled(None, None)

    def step_scaled(self, scale, found_inf):
        """step(); with (scale, found_inf) device scalars the gradients are first divided by `scale` and the update is
        skipped -- moments, parameters and step count untouched -- when any of them is inf/NaN (amp.GradScaler.step)."""
        from . import ops
        ops.flush_wgrad()                    # no-op unless a backward's deferred wgrad reduction is still pending
        params = dict(self.model.named_parameters())
        active = [n for n in self._names if params[n].grad is not None]
        if not active and not (self.frozen == "torch1" and self._ever):
            return
        _adopt_foreign_grads(params, active)
        if self.frozen == "torch1":
            # every parameter that has ever held a gradient steps (the frozen ones on the zeros zero_grad() left in the
            # flat gradient buffer); those stepping for the first time this round open a new step counter
            self._ever.update(active)
            grouped = set().union(*[set(k) for k in self._groups]) if self._groups else set()
            fresh = tuple(n for n in self._names if n in self._ever and n not in grouped)
            if fresh:
                if len(self._groups) >= self.MAX_GROUPS:
                    raise RuntimeError("FusedAdamW: too many distinct freeze patterns in one round")
                self._groups[fresh] = (len(self._groups), self.model.param_ranges(fresh))
            todo = list(self._groups.values())
        else:
            todo = [self._group_for(active)]
        P, G = self.model.flat_params, self.model.flat_grads
        if scale is not None:                 # all ranges first: an overflow anywhere skips every group's step
            for gi, ranges in todo:
                for s, e in ranges:
                    L.amp_unscale(G[s:e], scale, found_inf)
        for gi, ranges in todo:
            L.adamw_hyper(self.steps[gi:gi + 1], self.hyper[gi], self.lr_state, self.betas[0], self.betas[1], self.wd)
            if scale is not None:
                L.amp_guard(self.steps[gi:gi + 1], self.hyper[gi], found_inf)
            for s, e in ranges:
                L.adamw_step(P[s:e], G[s:e], self.m[s:e], self.v[s:e], self.hyper[gi], self.betas[0], self.betas[1],
                             self.eps, None if self.shadow is None else self.shadow[s:e])
        ops.bump_weights_epoch()            # raw-pointer write: invalidate the conv operand packs

    def advance_lr(self):
        """lr <- base_lr * (1 - iter/max_iterations)^0.9 with iter incremented first (flower_pCE_2D.py:150-157)."""
        L.lr_poly_advance(self.iter, self.lr_state, self.base_lr, float(self.max_iterations))


class FusedSGD:
    """``torch.optim.SGD(params, lr, momentum=0.9, weight_decay=1e-4)`` of the single-site trainer
    (/root/reference/code/Unet_pCE.py:88-89) over the flat parameter buffer, same device-resident LR / iteration state
    and interface as FusedAdamW (zero_grad / step / step_scaled / advance_lr)."""

    def __init__(self, model, lr, base_lr=None, max_iterations=None, momentum=0.9, weight_decay=1e-4):
        self.model, self.momentum, self.wd = model, momentum, weight_decay
        self.base_lr = lr if base_lr is None else base_lr
        self.max_iterations = max_iterations
        dev = model.flat_params.device
        self.buf = torch.zeros(model.flat_params.numel(), dtype=torch.float32, device=dev)
        self.lr_state = torch.tensor([lr], dtype=torch.float64, device=dev)
        self.iter = torch.zeros(1, dtype=torch.int32, device=dev)
        self._hyper = torch.zeros(4, dtype=torch.float32, device=dev)        # [0] < 0: skipped step (amp)
        self._step = torch.zeros(1, dtype=torch.int32, device=dev)
        self._names = [n_ for n_, _ in model.named_parameters()]

    def set_lr(self, lr: float, current_iter: int = 0):
        self.lr_state.fill_(lr)
        self.iter.fill_(int(current_iter))

    def zero_grad(self, set_to_none: bool = True):
        self.model.zero_grad(set_to_none)

    def step(self):
        self.step_scaled(None, None)

    def step_scaled(self, scale, found_inf):
        from . import ops
        ops.flush_wgrad()
        params = dict(self.model.named_parameters())
        active = [n for n in self._names if params[n].grad is not None]
        if not active:
            return
        _adopt_foreign_grads(params, active)     # the 3D ops hand autograd ordinary gradient tensors, not the flat sink
        ranges = self.model.param_ranges(tuple(active))
        P, G = self.model.flat_params, self.model.flat_grads
        skip = None
        if scale is not None:
            for s, e in ranges:
                L.amp_unscale(G[s:e], scale, found_inf)
            self._hyper.zero_()
            L.amp_guard(self._step, self._hyper, found_inf)
            skip = self._hyper
        for s, e in ranges:
            L.sgd_step(P[s:e], G[s:e], self.buf[s:e], self.lr_state, self.momentum, self.wd, skip)
        ops.bump_weights_epoch()

    def advance_lr(self):
        L.lr_poly_advance(self.iter, self.lr_state, self.base_lr, float(self.max_iterations))
