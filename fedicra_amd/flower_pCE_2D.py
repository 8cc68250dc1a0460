"""Local training procedure (pCE [+ LC] loss) with the reference's ``MyClient._train`` contract.

Mirrors /root/reference/code/flower_pCE_2D.py:42-181: fresh AdamW per round, epoch
pre-materialisation of the batch list, FedICRA freeze schedule and LC loss, poly LR after every
iteration, the same metrics dict.  MI355X-native differences:

  * forward / loss / backward / optimizer are the hand-written HIP kernels (fedicra_amd.ops);
  * the optimizer is one fused kernel over the flat parameter buffer with device-resident lr and
    step counters (fedicra_amd.optim), so the WHOLE iteration -- zero-grad, forward, CE,
    backward, AdamW, LR update -- is captured once per freeze pattern into a hipGraph
    (``args.use_graph``) and replayed; per-iteration losses stay on the device and are read back
    once per round instead of the reference's ``loss.item()`` sync every iteration (:152);
  * batches are staged into static device buffers (the graph's inputs).

Reference defect handled (LOG.md section 2): :117-118 unpacks UNet_LC's 8 outputs into 7 names; here the
logits are ``out[0]`` and the heat-maps ``out[6]``.
"""
from __future__ import annotations

import contextlib

import numpy as np
import torch

from . import fl, ops, streams
from .flower_common import BaseClient
from .optim import FusedAdamW


import os

_PROBE_BESIDE = os.environ.get("FEDICRA_PROBE_STREAM", "1") != "0"      # measurement switch (0: the LC forwards in line)


class _GraphStep:
    """One captured training iteration for one freeze pattern."""

    def __init__(self):
        self.graph = None
        self.loss = None
        self.loss_ce = None
        self.loss_lc = None
        self.logits = None


class MyClient(BaseClient):

    def __init__(self, args, model, trainloader, valloader, amp=False):
        super().__init__(args, model, trainloader, valloader)
        self.amp = amp
        if self.amp:                                         # flower_pCE_2D.py:46-48
            from .amp import GradScaler
            from .networks.unet import set_compute_dtype
            self.scaler = GradScaler()
            # autocast(enabled=True): fp16 storage / operands like the reference's torch.cuda.amp (amp.py); bf16 -- the same
            # MFMA rate on CDNA4, fp32's exponent range -- with args.amp_dtype = "bf16"
            set_compute_dtype(model.model, getattr(args, "amp_dtype", "fp16"))
        self.best_performance = 0.0
        self.use_graph = bool(getattr(args, "use_graph", False))
        self.optimizer = None
        self._steps = {}                                     # captured iterations of the current batch shape, by freeze phase
        self._shapes = {}                                    # batch shape -> (x buffer, y buffer, its captured iterations)
        self._xbuf = self._ybuf = None
        self.last_losses = []
        self.probe_beside = _PROBE_BESIDE                    # the LC forwards on a second stream beside the own forward
        # ... and their decoder half (BatchNorm statistics only: nothing the iteration reads) on beside the loss and the backward
        # pass, joined before the optimizer step (measurement switch: 0 = joined before the LC loss, as in round 3)
        self.probe_tail_beside = os.environ.get("FEDICRA_PROBE_TAIL", "1") != "0"
        self.aux_stats_only = os.environ.get("FEDICRA_AUX_STATS", "1") != "0"   # (measurement switch: 0 = the heads in full)
        self.ctx = ops.new_context()                         # arena / dropout counter a captured step bakes in
        self.stream = None                                   # set by whoever co-locates several clients on one GPU

    # ---------------------------------------------------------------------------------- helpers
    def _net(self):
        return self.model.model

    def _ensure_optimizer(self):
        if self.optimizer is None:
            self.optimizer = FusedAdamW(self._net(), lr=self.current_lr, base_lr=self.args.base_lr,
                                        max_iterations=self.args.max_iterations,
                                        frozen=getattr(self.args, "adamw_frozen", "torch2"))
            self.ctx.seed_offset = self.optimizer.iter       # ops.set_dropout_seed_offset, for this client's context
        return self.optimizer

    def _stage(self, sampled_batch):
        """Copy one batch into the static device buffers (graph inputs)."""
        dev = self._net().flat_state.device
        stager = self._stager()
        staged = stager is not None and stager.on_host(sampled_batch)
        x, y = stager.fetch(sampled_batch) if staged else (sampled_batch["image"], sampled_batch["label"])
        if self.args.img_class == "faz":
            x = x.unsqueeze(1)                               # flower_pCE_2D.py:77
        if self._xbuf is None or self._xbuf.shape != x.shape:
            # static buffers and captured steps are kept PER BATCH SHAPE: the short last batch of an epoch (drop_last=False,
            # as the reference's DataLoader) gets its own pair instead of invalidating the full-size graphs twice per epoch
            key = tuple(x.shape)
            if key not in self._shapes:
                self._shapes[key] = (torch.empty(x.shape, dtype=torch.float32, device=dev),
                                     torch.empty(y.shape, dtype=torch.uint8, device=dev), {})
            self._xbuf, self._ybuf, self._steps = self._shapes[key]
        self._xbuf.copy_(x, non_blocking=True)
        self._ybuf.copy_(y, non_blocking=True)
        if staged:
            stager.release()
        return self._xbuf, self._ybuf

    def _stager(self):
        """Host-resident batches (a DataLoader-like trainloader) are staged beside the compute stream (staging.py); the
        stager belongs to the model wrapper so that the ALA loop of set_weights shares it."""
        get = getattr(self.model, "batch_stager", None)
        return get() if callable(get) and self._net().flat_state.is_cuda else None

    def _prefetch_next(self, last=False, ala_expected=True):
        """Start the host -> device copy of the batch the NEXT iteration will take (flower_pCE_2D.py:66-73: same epoch
        list) while this one runs -- the next round's first iteration included.  At an epoch boundary the list is rebuilt
        first: only a list-like trainloader (whose next epoch is known) is looked into, otherwise that copy is serial.
        `ala_expected`: the set_weights that follows the round will run its ALA epoch (FedICRA and iter_global > 50,
        flower_common.py:524-526); when it will not, the epoch's batches are NOT sent ahead (they would cross PCIe for
        nothing and pin ring pairs -- ADVICE r4) and the next round's first batch is prefetched as after any iteration."""
        stager = self._stager()
        n_b = len(self.trainloader)
        if stager is None or not self.sampled_batches:
            return
        if last and ala_expected and self.args.strategy in ["FedICRA"] and isinstance(self.trainloader, (list, tuple)):
            # the round's last iteration is enqueued: what follows is set_weights with its ALA epoch over the whole loader
            # (flower_common.py:566-602) -- its batches cross PCIe now, beside the iterations the GPU still has queued, like a
            # DataLoader's workers run ahead (flower_pCE_2D.py:303-304); the first batch of the NEXT round is announced too
            self.model.next_batch_hint = self.sampled_batches[self.current_iter % n_b] if self.current_iter % n_b else self.trainloader[0]
            for b in self.trainloader[:max(0, stager.slots - 3)]:
                stager.prefetch(b)
            return
        if self.current_iter % n_b == 0:
            if isinstance(self.trainloader, (list, tuple)):
                stager.prefetch(self.trainloader[0])
            return
        stager.prefetch(self.sampled_batches[self.current_iter % n_b])

    def _set_freeze(self, i_iter):
        """flower_pCE_2D.py:84-101.  Returns a hashable pattern id."""
        if self.args.strategy not in ["FedICRA"]:
            return "all"
        local_keys = ["decoder.out_conv.weight", "decoder.out_conv.bias"]
        head_phase = i_iter < self.args.iters - self.args.rep_iters
        for name, param in self.model.named_parameters():
            param.requires_grad = (name.replace("model.", "") in local_keys) == head_phase
        self._head_only = head_phase                         # nothing below decoder.out_conv receives a gradient in this iteration
        return "head" if head_phase else "body"

    def _iteration(self, x, y, rec: _GraphStep):
        """zero-grad, forward, loss, backward, optimizer step, LR update -- all device work."""
        args = self.args
        opt = self.optimizer
        ops.begin_iteration(x.device)
        opt.zero_grad()
        batched, probe_stream, probe_tail, enc_done = None, None, None, None
        others = [c for c in range(args.min_num_clients) if c != args.cid]
        net = self.model.model
        can_batch = (args.strategy in ["FedICRA"] and hasattr(net, "probe_heatmaps") and x.is_cuda and ops.probe_ready()
                     and net.training)
        side = self.probe_beside and can_batch
        if side:
            # The K-1 no-grad LC forwards (:128-139) do not depend on the client's own forward (:106), only on the weights and
            # the batch: they run on a SECOND stream beside it -- the own forward's 12-image launches fill the gaps the batched
            # launches leave (measured: 91.0 -> 87.6 ms of training per round).  Order kept where it matters: the own forward is
            # enqueued first (dropout call counters), and every BatchNorm's running statistics take the own update before the
            # probe's (deferred to this stream after the join: ops._probe_finalize).  Fork / join are graph edges under capture.
            main = torch.cuda.current_stream()
            # (fedicra_amd/streams.py makes every role's stream in one place: never the stream this iteration runs / is captured on)
            probe_stream = streams.get("probe", x.device)
            assert probe_stream.cuda_stream != main.cuda_stream
            net._fi_refresh_packs(net.compute_dtype())       # packs the probe reads: ready BEFORE the fork
            fork = torch.cuda.Event()
            fork.record(main)
            # every BatchNorm's running statistics take the own forward's update first and the probe's K-1 after it: the
            # probe makes only its coefficient rows on its stream, its running-statistics updates are made HERE, on this
            # stream, after the join (ops._probe_finalize) -- no event between the two branches
            ops._ctx.probe_deferred = []
        deferred = None
        try:
            # this procedure reads the logits and the heat-map only (:117-139): on the LC models the auxiliary heads run for
            # their BatchNorm statistics alone (networks/unet._UNetLCBase.forward, aux="stats")
            out = self.model(x, aux="stats") if self.aux_stats_only and hasattr(net, "probe_heatmaps") else self.model(x)
            if side:
                probe_stream.wait_event(fork)
                enc_done = torch.cuda.Event() if self.probe_tail_beside else None
                with torch.cuda.stream(probe_stream), torch.no_grad():
                    batched = net.probe_heatmaps(x, others, enc_done=enc_done)
                if batched is None:
                    # the batched form does not apply to this model (probe_heatmaps' own preconditions): the second stream
                    # has already been forked into the capture by wait_event(fork) -- JOIN it before dropping it, a capture
                    # must not end with an unjoined stream
                    main.wait_stream(probe_stream)
                    probe_stream = None
        finally:
            deferred, ops._ctx.probe_deferred = ops._ctx.probe_deferred, None
        logits = out[0]
        loss_ce = ops.ce_loss(logits.permute(0, 2, 3, 1), y, args.num_classes)       # :124
        loss = loss_ce
        loss_lc = None
        if args.strategy in ["FedICRA"]:                                             # :128-139
            heatmaps = out[6]
            acc = 0
            if probe_stream is not None:
                if enc_done is not None:
                    # the heat-maps are complete once the probe's ENCODER is (the LC loss reads nothing else); its decoder only
                    # moves BatchNorm statistics and keeps running beside the loss and the backward pass -- joined before the
                    # optimizer touches the weights it reads
                    torch.cuda.current_stream().wait_event(enc_done)
                    probe_tail = probe_stream
                else:
                    torch.cuda.current_stream().wait_stream(probe_stream)
                    self._run_deferred(deferred)
                    deferred = None
            else:
                with torch.no_grad():                                                # all K-1 forwards as one batch
                    batched = self.model.model.probe_heatmaps(x, others) if hasattr(self.model.model, "probe_heatmaps") else None
            base = None
            if batched is not None and heatmaps[-1].is_cuda:
                base = batched[0]._base if batched[0]._base is not None else batched[0]
                if base.numel() != len(others) * heatmaps[-1].numel() or base.dtype != torch.float32 or not base.is_contiguous():
                    base = None
            if base is not None:
                # all K-1 heat-maps sit back to back in one tensor (probe_heatmaps returns views of it): the K-1 MSEs, their
                # mean, the sign and the alpha-weighted sum are ONE launch (fi_lc_loss_fwd)
                loss, loss_lc = ops.lc_total(loss_ce, heatmaps[-1], base.detach(), len(others), args.alpha)
            else:
                for k, other_client in enumerate(others):
                    if batched is not None:
                        other_map = batched[k]
                    else:
                        with torch.no_grad():
                            other_map = self.model(x, other_client, heatmap_only=True)[6][-1]   # nothing else of it is read
                    acc = acc + torch.nn.functional.mse_loss(heatmaps[-1], other_map.detach())
                loss_lc = -acc / (args.min_num_clients - 1)
                loss = torch.add(loss, loss_lc, alpha=args.alpha)
        if self.amp:                                         # :143-146
            self.scaler.scale(loss).backward()
            if probe_tail is not None:
                torch.cuda.current_stream().wait_stream(probe_tail)
            self._run_deferred(deferred)
            self.scaler.step(opt)
            self.scaler.update()
        else:
            loss.backward()
            if probe_tail is not None:
                torch.cuda.current_stream().wait_stream(probe_tail)
            self._run_deferred(deferred)
            opt.step()
        opt.advance_lr()
        rec.loss, rec.loss_ce, rec.loss_lc, rec.logits = loss.detach(), loss_ce.detach(), \
            (None if loss_lc is None else loss_lc.detach()), logits.detach()

    @staticmethod
    def _run_deferred(deferred):
        """The probe's running-statistics updates that were left to this stream (ops._probe_finalize): called once the probe
        stream has been joined."""
        ops.run_probe_deferred(deferred)

    # ---------------------------------------------------------------------------------- _train
    @contextlib.contextmanager
    def _scope(self):
        """This client's device context (ops._Context) and, when it shares the GPU with other clients, its HIP stream."""
        with contextlib.ExitStack() as st:
            st.enter_context(ops.use_context(self.ctx))
            if self.stream is not None:
                st.enter_context(torch.cuda.stream(self.stream))
            yield

    def _train(self, config):
        gen = self.train_steps(config)
        while True:
            try:
                next(gen)
            except StopIteration as done:
                return done.value

    def train_steps(self, config):
        """The local-training loop of flower_pCE_2D.py:51-181 as a generator: yields after ENQUEUEING each iteration (no
        host sync inside a round), returns ``(loss, metrics)`` like ``_train``.  A process that hosts several clients
        on one MI355X advances their generators round-robin, one HIP stream per client (`train_colocated` below)."""
        args = self.args
        with self._scope():
            self.model.train()
            opt = self._ensure_optimizer()
            opt.reset_round()                                    # fresh AdamW every round (:55)
            opt.set_lr(self.current_lr, self.current_iter)
            iters = config["iters"]
            dev = self._net().flat_state.device
            hist = torch.zeros((iters, 5), dtype=torch.float32, device=dev)   # loss, ce, lc, tree, crf
            n_b = len(self.trainloader)
        rec = None
        for i_iter in range(iters):
            with self._scope():
                if self.current_iter % n_b == 0:                 # :66-70 epoch pre-materialisation
                    self.sampled_batches.clear()
                    for sampled_batch in self.trainloader:
                        self.sampled_batches.append(sampled_batch)
                sampled_batch = self.sampled_batches[self.current_iter % n_b]
                x, y = self._stage(sampled_batch)
                pattern = self._set_freeze(i_iter)
                if opt.frozen == "torch1":
                    # which parameters step, and under which step COUNTERS, also depends on which ones have ever held a
                    # gradient and on how this round's fresh optimizer has grouped them so far (round 1: out_conv on
                    # counter 0, the rest on counter 1; later rounds: everything on counter 0 from the first head step on).
                    # A captured step bakes the counter addresses in, so the key carries the whole layout the step starts
                    # from -- which, with the freeze pattern, determines the layout it leaves
                    pattern = "{}/{}/{}".format(pattern, len(opt._ever), opt.layout_key())
                if self.use_graph:
                    rec = self._steps.get(pattern)
                    if rec is None:
                        rec = _GraphStep()
                        self._iteration(x, y, rec)               # first use of a pattern runs eagerly (real step)
                        self._steps[pattern] = rec
                    elif rec.graph is None:
                        torch.cuda.synchronize()
                        ops.reserve_graph_tables()               # pinned wgrad-reduce tables: not allocatable mid-capture
                        g = torch.cuda.CUDAGraph()
                        # thread_local: only this thread's calls are policed during the capture -- RCCL's watchdog
                        # thread polls events of its own and must not invalidate it (multi-GPU runs)
                        with torch.cuda.graph(g, stream=streams.get("capture"), capture_error_mode="thread_local"):     # records only; nothing executes
                            self._iteration(x, y, rec)
                        rec.graph = g
                        g.replay()
                        ops.bump_weights_epoch()                 # the replayed AdamW moved the weights
                    else:
                        rec.graph.replay()
                        ops.bump_weights_epoch()
                else:
                    rec = _GraphStep()
                    self._iteration(x, y, rec)
                hist[i_iter, 0] = rec.loss
                hist[i_iter, 1] = rec.loss_ce
                if rec.loss_lc is not None:
                    hist[i_iter, 2] = rec.loss_lc
                if getattr(rec, "loss_tree", None) is not None:      # the tree-energy procedure's extra terms
                    hist[i_iter, 3] = rec.loss_tree
                    hist[i_iter, 4] = rec.loss_crf
                self.current_iter += 1
                lr_ = args.base_lr * (1.0 - self.current_iter / args.max_iterations) ** 0.9      # :154 (host mirror)
                self.current_lr = lr_
                self._prefetch_next(last=(i_iter == iters - 1), ala_expected=config.get("iter_global", 51) > 50)
            yield i_iter
        with self._scope():
            return self._round_result(hist, x, y, rec)

    def _round_result(self, hist, x, y, rec):
        args = self.args
        h = hist.cpu().numpy()                               # ONE sync per round
        self.last_losses = h[:, 0].tolist()
        self.last_terms = [(float(r[0]), float(r[1]), float(r[3]), float(r[4]), float(r[2])) for r in h]   # loss, ce, tree, crf, lc
        # ---- pack general metrics (:160-175)
        image = x[1, :, :, :]
        image = (image - image.min()) / (image.max() - image.min())
        pred = torch.argmax(torch.softmax(rec.logits, dim=1), dim=1, keepdim=True)
        pred = pred[1, ...] * 50
        labs = y[1, ...].unsqueeze(0) * 50
        if args.img_class in ("odoc", "polyp"):
            pred, labs = pred.repeat(3, 1, 1), labs.repeat(3, 1, 1)
        metrics_ = {
            "client_{}_lr".format(self.cid): self.current_lr,
            "client_{}_total_loss".format(self.cid): float(h[-1, 0]),
            "client_{}_loss_ce".format(self.cid): float(h[-1, 1]),
            "client_{}_Image".format(self.cid): fl.ndarray_to_bytes(image.cpu().numpy()),
            "client_{}_Prediction".format(self.cid): fl.ndarray_to_bytes(pred.cpu().numpy()),
            "client_{}_GroundTruth".format(self.cid): fl.ndarray_to_bytes(labs.cpu().numpy()),
        }
        if args.strategy in ["FedICRA"]:
            metrics_["client_{}_loss_lc".format(self.cid)] = float(h[-1, 2])
        return float(h[-1, 0]), metrics_


def train_colocated(clients, configs):
    """Local training of several clients hosted by ONE process on ONE MI355X: each client owns a HIP stream and a device
    context, the host enqueues iteration i of every client before iteration i+1 of any, so their captured steps overlap
    on the GPU (a single 12x256^2 step is a chain of ~230 short launches that leaves most of the 256 CUs idle at any
    instant; two interleaved chains fill them: +40 % aggregate images/s measured).  The reference runs co-located clients as
    separate processes on the same CUDA device (flower_runner.py:100-102 maps client k to GPU k mod #gpus).
    -> [(loss, metrics)] in client order, as ``_train`` returns them."""
    main = torch.cuda.current_stream()
    for c in clients:
        if c.stream is None:
            c.stream = torch.cuda.Stream()
        c.stream.wait_stream(main)                           # the weights just loaded on the caller's stream
    gens = [c.train_steps(cfg) for c, cfg in zip(clients, configs)]
    out = [None] * len(clients)
    live = list(range(len(clients)))
    while live:
        for k in list(live):
            try:
                next(gens[k])
            except StopIteration as done:
                out[k] = done.value
                live.remove(k)
    for c in clients:
        main.wait_stream(c.stream)                           # aggregation reads the trained weights on the caller's stream
    return out
