"""Client / model-wrapper / strategy layer with the reference's API, MI355X-native underneath.

Mirrors /root/reference/code/flower_common.py for the hot path (SURVEY.md section 8-a1..a4, a16,
a17): ``BaseClient`` (get_parameters / get_properties / fit / evaluate), ``MyModel``
(get_weights / set_weights incl. the FedICRA adaptive local aggregation), ``evaluate``,
``aggregate`` / ``FedAvg`` / ``FedICRA`` and the metric aggregation functions -- same names,
argument meaning and quirks -- but

  * weights stay resident in HBM as ONE flat fp32 buffer (fedicra_amd/flat.py).  ``get_weights``
    still returns the reference's list of numpy arrays (wire format, state_dict order) for a
    Flower/gRPC server; the RCCL path exchanges ``DeviceWeights`` (flat tensors) with no host hop;
  * the ALA loop (flower_common.py:566-602) runs one fused element-wise kernel per batch over the
    contiguous decoder range instead of 2x42 tensor sweeps;
  * Dice bookkeeping (val_2D.py:9-22) is a device reduction; validation images go through the
    network as one batch (eval-mode BN makes images independent).

Third-party arithmetic restated (not vendored by the reference): flwr 1.0.0 ``aggregate``,
medpy 0.4.0 ``metric.binary.*`` -- see oracle/fed_ref.py, oracle/losses_ref.py.
"""
from __future__ import annotations

import copy
import os
import timeit
from collections import OrderedDict
from dataclasses import dataclass
from functools import reduce
from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn as nn

from . import _lib as L
from . import fl, ops, streams
from .flat import _as_view

VAL_METRICS = ["dice", "hd95", "recall", "precision", "jc", "specificity", "ravd"]     # flower_common.py:121
PERSONALIZED_FL = ["FedICRA"]
CENTRALIZED_FL = ["FedAvg", "FedAdagrad", "FedAdam", "FedYogi"]
ALA_KEYS = ["out_conv", "up4", "up3", "up2", "up1"]                                      # flower_common.py:506


@dataclass
class DeviceWeights:
    """Zero-copy form of the wire payload for the RCCL path: the flat fp32 state (parameters + BN
    running statistics) and the int64 num_batches_tracked counters, both on the device."""
    state: torch.Tensor
    counters: torch.Tensor


# ------------------------------------------------------------------------------ evaluation
def dice_table(logits_nchw_view, labels_u8, classes):
    """Per-image integer counts {|P&G|, |P|, |G|} for classes 1..C-1 (val_2D.py:66-74) -> int64 [n, C-1, 3]."""
    lg = logits_nchw_view.permute(0, 2, 3, 1)
    if not lg.is_contiguous() or lg.dtype != torch.float32:
        lg = lg.contiguous().float()
    n = lg.shape[0]
    counts = torch.zeros((n, classes - 1, 3), dtype=torch.int64, device=lg.device)
    for i in range(n):
        L.dice_counts(lg[i], labels_u8[i], counts[i])
    return counts


def hd95_table(logits_nchw_view, labels_u8, classes):
    """medpy.metric.binary.hd95 (val_2D.py:14) per image and foreground class -> float64 [n, C-1].

    The border extraction and the nearest-border search run on the device (fi_seg_borders / fi_surface_distances,
    exact); numpy takes the 95th percentile of the two directed distance sets, as medpy does.  NaN where the prediction
    or the ground truth has no pixel of the class (medpy raises there; the reference only guards the empty prediction)."""
    lg = logits_nchw_view.permute(0, 2, 3, 1)
    if not lg.is_contiguous() or lg.dtype != torch.float32:
        lg = lg.contiguous().float()
    n, H, W, _ = lg.shape
    dev = lg.device
    out = np.full((n, classes - 1), np.nan)
    plist = torch.empty(H * W, dtype=torch.int32, device=dev)
    glist = torch.empty(H * W, dtype=torch.int32, device=dev)
    d_pg = torch.empty(H * W, dtype=torch.float64, device=dev)
    d_gp = torch.empty(H * W, dtype=torch.float64, device=dev)
    counts = torch.zeros(2, dtype=torch.int32, device=dev)
    for i in range(n):
        for k in range(1, classes):
            counts.zero_()
            L.seg_borders(lg[i], labels_u8[i], k, plist, glist, counts)
            L.surface_distances(plist, glist, counts, 0, 1, W, d_pg)
            L.surface_distances(glist, plist, counts, 1, 0, W, d_gp)
            n_p, n_g = (int(v) for v in counts.cpu())
            if n_p == 0 or n_g == 0:
                continue
            sds = np.concatenate([d_pg[:n_p].cpu().numpy(), d_gp[:n_g].cpu().numpy()])
            out[i, k - 1] = np.percentile(sds, 95)
    return out


def metrics_from_counts(tp, npred, ngt, total, hd95=float("nan")):
    """The medpy metrics of val_2D.py:13-19 for ONE image/class from its integer counts (+ hd95 from hd95_table)."""
    if npred == 0:                                        # val_2D.py:12,21-22
        return [0.0] * 7
    fp, fn = npred - tp, ngt - tp
    tn = total - tp - fp - fn
    dice = 2.0 * tp / float(npred + ngt)
    recall = tp / float(tp + fn) if (tp + fn) else 0.0
    precision = tp / float(tp + fp) if (tp + fp) else 0.0
    union = npred + ngt - tp
    jc = tp / float(union) if union else 0.0
    spec = tn / float(tn + fp) if (tn + fp) else 0.0
    ravd = (npred - ngt) / float(ngt) if ngt else float("nan")
    return [dice, float(hd95), recall, precision, jc, spec, ravd]


EVAL_CHUNK = 16        # validation images per forward (the reference: one, val_2D.py:25-74)


def evaluate(args, model, dataloader, amp=False):
    """flower_common.py:122-136.  The validation forward runs in chunks of EVAL_CHUNK images: eval-mode BatchNorm makes the
    images independent, so the metrics are those of the reference's one-image-at-a-time loop, while the activations of a
    chunk (the full-resolution 512-channel head of unet_lc_multihead is ~0.6 GB per 384^2 image) stay bounded whatever
    the size of the split."""
    dev = next(model.parameters()).device
    was_training = model.training
    model.eval()                                          # val_2D.py:40
    ncls = args.num_classes
    metric_list = np.zeros((ncls - 1, len(VAL_METRICS)))
    seen = 0
    xs, ys, held = [], [], 0

    def flush():
        nonlocal metric_list, seen, xs, ys, held
        if not xs:
            return
        x, y = torch.cat(xs).float().to(dev), torch.cat(ys).to(torch.uint8).to(dev)
        xs, ys, held = [], [], 0
        with torch.no_grad():
            logits = model(x)[0]
        counts = dice_table(logits, y, ncls).cpu().numpy()
        hd = hd95_table(logits, y, ncls)
        total = int(y.shape[-1] * y.shape[-2])
        for i in range(counts.shape[0]):
            metric_list += np.array([metrics_from_counts(*map(int, counts[i, c]), total, hd[i, c]) for c in range(ncls - 1)])
        seen += counts.shape[0]

    for b in dataloader:
        x, y = b["image"], b["label"]
        x = x.reshape((-1,) + tuple(x.shape[-2:])) if args.img_class == "faz" else x.reshape((-1,) + tuple(x.shape[-3:]))
        xs.append(x.unsqueeze(1) if args.img_class == "faz" else x)
        ys.append(y.reshape((-1,) + tuple(y.shape[-2:])))
        held += xs[-1].shape[0]
        if held >= EVAL_CHUNK:
            flush()
    flush()
    model.train(was_training) if was_training else None
    n_images = len(dataloader.dataset) if hasattr(dataloader, "dataset") else seen
    metric_list = metric_list / n_images
    metrics_ = {}
    for class_i in range(ncls - 1):
        for mi, name in enumerate(VAL_METRICS):
            metrics_["val_{}_{}".format(class_i + 1, name)] = metric_list[class_i, mi]
    for mi, name in enumerate(VAL_METRICS):
        metrics_["val_mean_{}".format(name)] = np.mean(metric_list, axis=0)[mi]
    return metrics_


# ------------------------------------------------------------------------------ model wrapper
class MyModel(nn.Module):
    """flower_common.py:458-633."""

    def __init__(self, args, model, trainloader, valloader):
        super().__init__()
        self.args = args
        self.model = model
        self.trainloader = trainloader
        self.valloader = valloader
        self.amp = (getattr(args, "amp", 0) == 1)
        if self.amp:                                          # flower_common.py:466-468
            from .amp import GradScaler
            self.scaler = GradScaler()
        if self.args.strategy in ["FedICRA"]:
            self.start_phase = True
        self.fedaa_weights = None
        self.ala_epoch_losses: List[float] = []

    def forward(self, x, emb_idx=None, heatmap_only=False, aux=True):
        if emb_idx is None:
            return self.model(x) if aux is True else self.model(x, aux=aux)
        if heatmap_only:                                  # LC models only (networks/unet._UNetLCBase.forward)
            return self.model(x, emb_idx, heatmap_only=True)
        return self.model(x, emb_idx)

    # -- weights I/O ----------------------------------------------------------------------------
    def get_weights(self, config=None):
        """List of numpy arrays in state_dict order (flower_common.py:488-489) -- the Flower wire format."""
        net = self.model
        if not hasattr(net, "_fi_offsets"):
            return [val.detach().cpu().numpy() for _, val in net.state_dict().items()]
        # the reference pays one D2H sync per array (136-160 of them); the flat store needs two copies in total
        host, cnt = net.flat_state.detach().cpu(), net.flat_counters.cpu()
        out, ci = [], 0
        for k, v in net.state_dict().items():
            if v.is_floating_point():
                out.append(_as_view(host, net._fi_offsets[k], v).numpy())
            else:
                out.append(cnt[ci].numpy())
                ci += 1
        return out

    def get_device_weights(self) -> DeviceWeights:
        return DeviceWeights(self.model.flat_state, self.model.flat_counters)

    def _load_global(self, weights):
        if isinstance(weights, DeviceWeights):
            self.model.flat_state.copy_(weights.state)
            self.model.flat_counters.copy_(weights.counters)
        elif not hasattr(self.model, "_fi_offsets"):
            sd = OrderedDict((k, torch.tensor(v)) for k, v in zip(self.model.state_dict().keys(), weights))
            self.model.load_state_dict(sd, strict=False)      # int64 buffers: float64 -> truncation (quirk 6)
        else:
            # same pairing and conversions as load_state_dict(zip(keys, tensors), strict=False), assembled on the host
            # and uploaded with two copies instead of one per key
            net = self.model
            host = torch.empty(net.flat_state.shape, dtype=torch.float32).copy_(net.flat_state)
            cnt = net.flat_counters.cpu().clone()
            ci = 0
            for (k, v), arr in zip(net.state_dict().items(), weights):
                t = torch.as_tensor(np.asarray(arr))
                if tuple(t.shape) != tuple(v.shape):
                    raise RuntimeError("size mismatch for {}: copying a param with shape {} from checkpoint, the shape in "
                                       "current model is {}.".format(k, tuple(t.shape), tuple(v.shape)))
                if v.is_floating_point():
                    _as_view(host, net._fi_offsets[k], v).copy_(t)
                else:
                    cnt[ci] = t.to(torch.int64)               # float64 -> int64 truncation (quirk 6)
                    ci += 1
            net.flat_state.copy_(host)
            net.flat_counters.copy_(cnt)
            ops.bump_weights_epoch()

    def batch_stager(self):
        """Host -> device staging beside the compute stream (staging.py), shared by local training and the ALA loop."""
        st = self.__dict__.get("_stager")
        if st is None:
            from .staging import BatchStager
            st = self.__dict__["_stager"] = BatchStager(self.model.flat_state.device)
        return st

    def stage_ahead(self, sampled_batch):
        """Start the host -> device copy of a batch that set_weights / _train will ask for next (round drivers call this while
        the aggregation collective runs: 'the next round's data staging' it is overlapped with)."""
        if self.model.flat_state.is_cuda and sampled_batch is not None and sampled_batch["image"].device.type == "cpu":
            self.batch_stager().prefetch(sampled_batch)

    def _batch(self, sampled_batch):
        """(image, label, staged) on the device.  flower_common.py:568-573.  A host batch comes out of the staging pair
        (prefetched on the side stream when the caller looked ahead); `staged` tells the caller to release() it."""
        dev = self.model.flat_state.device
        staged = dev.type == "cuda" and sampled_batch["image"].device.type == "cpu"
        if staged:
            x, y = self.batch_stager().fetch(sampled_batch)
        else:
            x, y = sampled_batch["image"].to(dev), sampled_batch["label"].to(dev)
        if self.args.img_class == "faz":
            x = x.unsqueeze(1)
        return x, y, staged

    def _ala_model(self, net, local_keys):
        """The reference deep-copies the model on every set_weights (:498,503); here ONE copy is kept and its flat state
        refreshed from the live model (a single device copy), so that the captured ALA iteration -- which bakes in the
        copy's buffers, the mixing weights, the old-local snapshot and the staging buffers -- survives across rounds."""
        st = getattr(self, "_ala", None)
        if st is None or st["src"] is not net:
            temp = copy.deepcopy(net)
            for n, p in temp.named_parameters():              # :542-546
                p.requires_grad = n in local_keys
            s, e = net.param_ranges(local_keys)[0]
            dev = net.flat_params.device
            ctx = ops.new_context()
            st = {"src": net, "temp": temp, "w": torch.ones(e - s, dtype=torch.float32, device=dev),
                  "old": torch.empty_like(net.flat_params), "iter": torch.zeros(1, dtype=torch.int32, device=dev),
                  "ctx": ctx, "x": None, "y": None, "graph": None, "warm": False, "loss": None}
            ctx.seed_offset = st["iter"]
            self.__dict__["_ala"] = st                        # not a submodule: keep it out of state_dict()
        temp = st["temp"]
        temp.flat_state.copy_(net.flat_state)                 # parameters AND BatchNorm running statistics
        temp.flat_counters.copy_(net.flat_counters)
        temp.train(net.training)
        return temp

    # ---- the ALA epoch with the encoder hoisted out of the batch loop ------------------------------------------------
    def _ala_batched_ok(self, temp, use_graph):
        """The ALA loop trains nothing but the decoder of the copy (flower_common.py:542-546), so inside an epoch its encoder
        sees the same weights for every batch: its forward over ALL batches of the epoch can run as ONE batched no-grad
        pass of statistics groups (group = batch: per-batch BatchNorm statistics, per-batch dropout draws; the fused
        forms of the LC forwards, networks.unet.LCEncoder._probe_batches) instead of len(trainloader) passes of 12 images
        that leave most of the chip idle.  Applies with the device RNG (parity runs that feed host masks keep the per-batch
        loop: the reference's draw order), without amp, on a list-like loader of equally shaped batches."""
        if os.environ.get("FEDICRA_ALA_BATCHED", "1") == "0" or not getattr(self, "ala_batched", True):
            return False
        if ops._mask_provider is not None or self.amp or not getattr(self, "ala_skip_aux", True):
            return False
        enc = getattr(temp, "encoder", None)
        if enc is None or not hasattr(enc, "_probe_batches") or getattr(enc, "n_pcs", 0) != 1 or not temp.training:
            return False
        if not isinstance(self.trainloader, (list, tuple)) or not (2 <= len(self.trainloader) <= 16):
            return False
        if not temp.flat_params.is_cuda or not hasattr(temp.decoder, "_run"):
            return False
        dec = temp.decoder
        if not all(getattr(b, "bilinear", True) for b in (dec.up1, dec.up2, dec.up3, dec.up4)):
            return False
        dtp = temp.compute_dtype()
        if any(c % (4 if dtp == torch.float32 else 8) for c in enc.ft_chns):
            return False
        b0 = self.trainloader[0]
        return all(b["image"].shape == b0["image"].shape and b["label"].shape == b0["label"].shape for b in self.trainloader)

    def _ala_epoch_batched(self, st, temp, tp, tg, old_local, glob, s, e, w, eta, ncls, use_graph):
        """One ALA epoch (flower_common.py:566-602) -> the last batch's loss (device scalar).  Every batch is staged into ONE
        static input; the epoch -- encoder pass over all of them, then per batch: decoder forward, pCE, decoder-only backward,
        mixing update -- is captured into one hipGraph at its second use and replayed from then on."""
        batches = list(self.trainloader)
        G = len(batches)
        dev = temp.flat_params.device
        ep = st.get("ep")
        xs0 = batches[0]["image"]
        n = xs0.shape[0]
        xshape = (G * n, 1) + tuple(xs0.shape[1:]) if self.args.img_class == "faz" else (G * n,) + tuple(xs0.shape[1:])
        if ep is None or ep["x"].shape != xshape:
            ep = st["ep"] = {"x": torch.empty(xshape, dtype=torch.float32, device=dev),
                             "y": torch.empty((G * n,) + tuple(batches[0]["label"].shape[1:]), dtype=torch.uint8, device=dev),
                             "graph": None, "warm": False, "loss": None}
        for b, sampled_batch in enumerate(batches):          # staging: batch b + 1 crosses PCIe while batch b is copied in
            x, y, staged = self._batch(sampled_batch)
            ep["x"][b * n:(b + 1) * n].copy_(x, non_blocking=True)
            ep["y"][b * n:(b + 1) * n].copy_(y, non_blocking=True)
            if staged:
                self.batch_stager().release()
                if b + 1 < G:
                    self.batch_stager().prefetch(batches[b + 1])
        hint = getattr(self, "next_batch_hint", None)       # the batch the next training round starts with (set by the client)
        if hint is not None and dev.type == "cuda" and hint["image"].device.type == "cpu":
            self.batch_stager().prefetch(hint)

        dtp = temp.compute_dtype()

        def epoch():
            ops.begin_iteration(dev)
            temp._fi_refresh_packs(dtp)
            with torch.no_grad():
                feats = temp.encoder._probe_batches(temp._in(ep["x"]), G)
            loss = None
            for b in range(G):
                temp.zero_grad()
                temp._fi_refresh_packs(dtp)                  # the decoder's operands follow the mixing update of batch b - 1
                o = temp.decoder._run([t[b * n:(b + 1) * n] for t in feats], False, False)
                loss = ops.ce_loss(o[0], ep["y"][b * n:(b + 1) * n], ncls)
                loss.backward()
                ops.flush_wgrad()
                L.ala_update(w, tp[s:e], tg[s:e], old_local[s:e], glob[s:e], eta, None)
                ops.bump_weights_epoch()
            st["iter"].add_(G)                               # fresh dropout masks for every batch of the next epoch
            return loss

        if not use_graph:
            return epoch()
        if not ep["warm"]:
            ep["loss"] = epoch()
            ep["warm"] = True
        elif ep["graph"] is None:
            torch.cuda.synchronize()
            ops.reserve_graph_tables(G + 2)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=streams.get("capture"), capture_error_mode="thread_local"):
                ep["loss"] = epoch()
            ep["graph"] = g
            g.replay()
        else:
            ep["graph"].replay()
        ops.bump_weights_epoch()
        return ep["loss"]

    def set_weights(self, weights, config):
        if self.args.strategy not in ["FedICRA"]:
            self._load_global(weights)                       # flower_common.py:627-633
            return
        # ---------------- FedICRA adaptive local aggregation, flower_common.py:494-624 -------------
        eta, num_pre_loss, threshold = 1.0, 10, 0.1
        net = self.model
        old_local = net.flat_params.clone()                   # "server_model" deepcopy = OLD LOCAL weights (quirk 4)
        self._load_global(weights)                            # self.model now holds the GLOBAL weights
        mark = self.__dict__.get("_timing_mark")              # bench.py's round split: an event after the global load
        if mark is not None:
            mark("loaded")
        glob = net.flat_params
        first = next(iter(net.parameters()))
        n0, o0 = first._fi_off, first.numel()
        if torch.sum(old_local[n0:n0 + o0] - glob[n0:n0 + o0]) == 0:      # :520-522 (a sum test, as written)
            return
        if config["iter_global"] <= 50:                       # :524-526
            if getattr(self, "verbose", True):
                print("skip", config)
            return
        local_keys = [n for n, _ in net.named_parameters() if any(k in n for k in ALA_KEYS)]
        ranges = net.param_ranges(local_keys)
        assert len(ranges) == 1, "decoder parameters are contiguous in the flat buffer"
        s, e = ranges[0]
        temp = self._ala_model(net, local_keys)               # :503 deepcopy + :542-546 requires_grad, cached across calls
        st = self._ala
        w = st["w"]
        w.fill_(1.0)                                          # re-initialised on every call (quirk 3)
        st["old"].copy_(old_local)
        old_local = st["old"]                                 # static address: the captured iteration reads it
        tp, tg = temp.flat_params, temp.flat_grads
        tp[s:e].copy_(old_local[s:e])                         # temp = global + (local - global) * 1
        losses, count = [], 0
        ncls = self.args.num_classes
        use_graph = bool(getattr(self.args, "use_graph", False)) and ops._mask_provider is None

        def iteration(x, y):
            """One ALA batch (:566-602): device work only, hipGraph-capturable."""
            ops.begin_iteration(x.device)
            temp.zero_grad()
            # the copy's auxiliary heads feed nothing the loop reads (only [0]) and its BatchNorm statistics are discarded with
            # it (:623-624 copies the 42 decoder tensors back, nothing else): skip them -- unless masks come from the host
            # generator (parity runs), where the skipped Dropout2d would shift every later draw of the run
            lean = getattr(self, "ala_skip_aux", True) and ops._mask_provider is None and "aux" in getattr(temp.forward, "__code__", type("c", (), {"co_varnames": ()})).co_varnames
            out = (temp(x, aux=False) if lean else temp(x))[0]
            loss = ops.ce_loss(out.permute(0, 2, 3, 1), y, ncls)
            skip = None
            if self.amp:                                      # :576-584: scaled backward, unscale (the lr = 0 step), update
                self.scaler.scale(loss).backward()
                ops.flush_wgrad()
                skip = self.scaler.unscale_range(tg[s:e])
            else:
                loss.backward()
                ops.flush_wgrad()
            # w <- clamp(w - eta*grad*(local-global), 0, 1); temp <- global + (local-global)*w   (one kernel); an
            # overflowed amp batch moves nothing (the reference would pour its inf/NaN gradients into w)
            L.ala_update(w, tp[s:e], tg[s:e], old_local[s:e], glob[s:e], eta, skip)
            if self.amp:
                self.scaler.update()
            st["iter"].add_(1)                                # fresh dropout masks for the next batch, replay included
            return loss

        fast = self._ala_batched_ok(temp, use_graph)
        with ops.use_context(st["ctx"]):
            while True:
                loss = None
                if fast:
                    # the frozen encoder of ALL batches of the epoch as one batched pass, then decoder forward / backward /
                    # mixing update per batch (_ala_epoch_batched)
                    loss = self._ala_epoch_batched(st, temp, tp, tg, old_local, glob, s, e, w, eta, ncls, use_graph)
                batches = iter(() if fast else self.trainloader)      # :566-602, one batch of look-ahead for the H2D staging
                upcoming = next(batches, None)
                while upcoming is not None:
                    sampled_batch, upcoming = upcoming, next(batches, None)
                    x, y, staged = self._batch(sampled_batch)
                    if not use_graph:
                        if staged:                            # eager path: take a private copy, free the staging pair
                            x, y = x.clone(), y.clone()
                            self.batch_stager().release()
                            self.batch_stager().prefetch(upcoming)
                        loss = iteration(x, y)
                        ops.bump_weights_epoch()              # temp's weights were rewritten through raw pointers
                        continue
                    if st["x"] is None or st["x"].shape != x.shape:
                        # static buffers + captured iteration PER BATCH SHAPE (a short last batch keeps its own pair)
                        shapes = st.setdefault("shapes", {})
                        if st["x"] is not None:
                            shapes[tuple(st["x"].shape)] = (st["x"], st["y"], st["graph"], st["warm"], st.get("loss"))
                        if tuple(x.shape) in shapes:
                            st["x"], st["y"], st["graph"], st["warm"], st["loss"] = shapes[tuple(x.shape)]
                        else:
                            st["x"], st["y"] = torch.empty_like(x), torch.empty_like(y)
                            st["graph"], st["warm"] = None, False
                    st["x"].copy_(x, non_blocking=True)
                    st["y"].copy_(y, non_blocking=True)
                    if staged:
                        self.batch_stager().release()
                        self.batch_stager().prefetch(upcoming)         # lands while this batch computes
                    if not st["warm"]:                        # first batch ever: eager (allocations, operand packs)
                        st["loss"] = iteration(st["x"], st["y"])
                        st["warm"] = True
                    elif st["graph"] is None:
                        torch.cuda.synchronize()
                        ops.reserve_graph_tables()
                        g = torch.cuda.CUDAGraph()
                        with torch.cuda.graph(g, stream=streams.get("capture"), capture_error_mode="thread_local"):
                            st["loss"] = iteration(st["x"], st["y"])
                        st["graph"] = g
                        g.replay()
                    else:
                        st["graph"].replay()
                    ops.bump_weights_epoch()
                    loss = st["loss"]
                losses.append(float(loss.item()))             # one host sync per ALA epoch (the reference: per batch)
                count += 1
                if getattr(self, "verbose", True):             # the reference prints this line every epoch (:606-607)
                    print("Client:", self.args.cid, "\tStd:", np.std(losses[-num_pre_loss:]), "\tALA epochs:", count,
                          self.start_phase)
                if not self.start_phase:                      # :611-612
                    break
                if len(losses) > num_pre_loss and np.std(losses[-num_pre_loss:]) < threshold:      # :615
                    break
        self.start_phase = False
        glob[s:e].copy_(tp[s:e])                              # :623-624
        self.fedaa_weights = w.clone()                        # w itself is the captured iteration's static buffer
        self.ala_epoch_losses = losses


# ------------------------------------------------------------------------------ client
class BaseClient:
    """flower_common.py:37-118 (fl.client.Client protocol)."""

    def __init__(self, args, model, trainloader, valloader):
        self.args = args
        self.cid = args.cid
        self.model = model
        self.trainloader = trainloader
        self.valloader = valloader
        self.current_iter = 0
        self.current_lr = self.args.base_lr
        self.sampled_batches = []
        self.properties = {"cid": self.cid}
        self.best_performance = 0.0
        self.amp = False

    def get_parameters(self, ins):
        weights = self.model.get_weights(getattr(ins, "config", None))
        return fl.GetParametersRes(status=fl.Status("OK", "Success"), parameters=fl.ndarrays_to_parameters(weights))

    def get_properties(self, ins):
        return fl.GetPropertiesRes(status=fl.Status("OK", "Success"), properties=self.properties)

    def fit(self, ins):
        weights = ins.parameters if isinstance(ins.parameters, DeviceWeights) else fl.parameters_to_ndarrays(ins.parameters)
        config = ins.config
        fit_begin = timeit.default_timer()
        self.model.set_weights(weights, config)
        loss, metrics_ = self._train(config)
        if isinstance(ins.parameters, DeviceWeights):
            params_prime = self.model.get_device_weights()
        else:
            params_prime = fl.ndarrays_to_parameters(self.model.get_weights(config))
        metrics_["fit_duration"] = timeit.default_timer() - fit_begin
        return fl.FitRes(status=fl.Status("OK", "Success"), parameters=params_prime,
                         num_examples=len(self.trainloader), metrics=metrics_)       # = number of BATCHES (quirk 5)

    def evaluate(self, ins):
        weights = ins.parameters if isinstance(ins.parameters, DeviceWeights) else fl.parameters_to_ndarrays(ins.parameters)
        self.model.set_weights(weights, ins.config)          # FedICRA: re-runs an ALA epoch (quirk 9)
        loss, metrics_ = self._validate(ins.config)
        return fl.EvaluateRes(status=fl.Status("OK", "Success"), loss=loss, num_examples=len(self.valloader),
                              metrics=metrics_)

    def _train(self, config):
        raise NotImplementedError

    def _validate(self, config):
        self.model.eval()
        val_metrics = evaluate(self.args, self.model, self.valloader, self.amp)
        if val_metrics["val_mean_dice"] > self.best_performance:
            self.best_performance = val_metrics["val_mean_dice"]
            snap = getattr(self.args, "snapshot_path", None)
            if snap:                                          # flower_common.py:106-113 checkpoint names
                sd = {k: v.detach().cpu().contiguous() for k, v in self.model.model.state_dict().items()}
                torch.save(sd, os.path.join(snap, "client_{}_async_iter_{}_dice_{}.pth".format(
                    self.cid, self.current_iter, round(self.best_performance, 4))))
                torch.save(sd, os.path.join(snap, "client_{}_async_{}_best_model.pth".format(self.cid, self.args.model)))
        return 0.0, {"client_{}_{}".format(self.cid, k): v for k, v in val_metrics.items()}


# ------------------------------------------------------------------------------ server-side arithmetic
def aggregate(results: Sequence[Tuple[Sequence[np.ndarray], int]]) -> List[np.ndarray]:
    """flwr 1.0.0 ``aggregate`` on host arrays (compat path for a Flower server)."""
    total = sum(n for _, n in results)
    weighted = [[layer * n for layer in w] for w, n in results]
    return [reduce(np.add, layers) / total for layers in zip(*weighted)]


def aggregate_device(results: Sequence[Tuple[DeviceWeights, int]]) -> DeviceWeights:
    """The same weighted mean over K flat device states held by ONE process (virtual clients):
    acc = acc + (w_k * n_k) left to right, then / total -- numpy's order and roundings (fi_axpy / fi_scale)."""
    total = sum(n for _, n in results)
    first = results[0][0].state
    acc = torch.empty_like(first)
    L.scale(first, acc, float(results[0][1]))
    for dw, n in results[1:]:
        L.axpy(acc, dw.state, float(n))
    L.scale(acc, acc, float(total), divide=True)
    cnt = sum(dw.counters * int(n) for dw, n in results)      # int64 sum, then true divide -> float64 -> truncate
    counters = (cnt.double() / total).to(torch.int64)
    return DeviceWeights(acc, counters)


class FedAvg:
    """The slice of flwr.server.strategy.FedAvg the reference uses (aggregate_fit, weights = num_examples)."""

    def __init__(self, fit_metrics_aggregation_fn=None, evaluate_metrics_aggregation_fn=None, accept_failures=False,
                 **kwargs):
        self.fit_metrics_aggregation_fn = fit_metrics_aggregation_fn
        self.evaluate_metrics_aggregation_fn = evaluate_metrics_aggregation_fn
        self.accept_failures = accept_failures
        self.kwargs = kwargs

    def aggregate_fit(self, server_round, results, failures):
        if not results:
            return None, {}
        if not self.accept_failures and failures:
            return None, {}
        res = [r for _, r in results]
        if isinstance(res[0].parameters, DeviceWeights):
            agg = aggregate_device([(r.parameters, r.num_examples) for r in res])
        else:
            agg = fl.ndarrays_to_parameters(aggregate([(fl.parameters_to_ndarrays(r.parameters), r.num_examples)
                                                       for r in res]))
        metrics = {}
        if self.fit_metrics_aggregation_fn:
            metrics = self.fit_metrics_aggregation_fn([(r.num_examples, r.metrics) for r in res])
        return agg, metrics

    def aggregate_evaluate(self, server_round, results, failures):
        if not results:
            return None, {}
        res = [r for _, r in results]
        total = sum(r.num_examples for r in res)
        loss = sum(r.num_examples * r.loss for r in res) / total
        metrics = {}
        if self.evaluate_metrics_aggregation_fn:
            metrics = self.evaluate_metrics_aggregation_fn([(r.num_examples, r.metrics) for r in res])
        return loss, metrics

    def __repr__(self):
        return f"FedAvg(accept_failures={self.accept_failures})"


class FedICRA(FedAvg):
    """flower_common.py:451-455: server side is plain FedAvg; the adaptive part lives in MyModel.set_weights."""

    def __repr__(self):
        return f"FedICRA(accept_failures={self.accept_failures})"


class FedOpt(FedAvg):
    """flwr 1.0.0 ``FedOpt`` family (Reddi et al., Adaptive Federated Optimization): FedAvg's weighted mean is treated as
    a pseudo-gradient ``delta = mean - current`` for a server-side Adagrad / Adam / Yogi step.  flwr is third-party and
    absent here: restated from its published source (strategy/fedopt.py, fedadagrad.py, fedadam.py, fedyogi.py) --
    *parity unpinned*.  Like flwr, ``initial_parameters`` is mandatory (the reference passes it only inside a commented
    block, flower_pCE_2D.py:353-363, so its own FedOpt runs fail at construction).  Host payloads (lists of numpy arrays)
    follow numpy's dtype rules array by array; ``DeviceWeights`` payloads run ``fi_fedopt_step`` on the flat state."""

    MODE = None
    DEFAULTS = {}

    def __init__(self, *, initial_parameters, eta=None, eta_l=None, beta_1=None, beta_2=None, tau=None, **kwargs):
        super().__init__(**kwargs)
        d = self.DEFAULTS
        self.eta = d["eta"] if eta is None else eta
        self.eta_l = d["eta_l"] if eta_l is None else eta_l
        self.beta_1 = d["beta_1"] if beta_1 is None else beta_1
        self.beta_2 = d["beta_2"] if beta_2 is None else beta_2
        self.tau = d["tau"] if tau is None else tau
        self.on_device = isinstance(initial_parameters, DeviceWeights)
        if self.on_device:
            self.current = DeviceWeights(initial_parameters.state.clone(), initial_parameters.counters.clone())
            self.cnt = initial_parameters.counters.clone()            # int64 like flwr's first current_weights
        else:
            self.current_weights = [np.array(a) for a in fl.parameters_to_ndarrays(initial_parameters)]
        self.m_t = self.v_t = None

    def _second_moment(self, v, d):
        raise NotImplementedError

    def aggregate_fit(self, server_round, results, failures):
        agg, metrics = super().aggregate_fit(server_round, results, failures)
        if agg is None:
            return None, {}
        if self.on_device:
            # the int64 counters: FedAvg hands back their float64 true-divide; the step runs in float64 (numpy's promotion)
            # and the clients truncate on load (quirk 6).  18 scalars: plain tensor arithmetic.
            total = sum(r.num_examples for _, r in results)
            avg = sum(r.parameters.counters.double() * int(r.num_examples) for _, r in results) / total
            return self.server_step(agg, avg), metrics
        fedavg_weights = fl.parameters_to_ndarrays(agg)
        delta_t = [x - y for x, y in zip(fedavg_weights, self.current_weights)]
        if not self.m_t:
            self.m_t = [np.zeros_like(x) for x in delta_t]
        self.m_t = [self.beta_1 * x + (1 - self.beta_1) * y for x, y in zip(self.m_t, delta_t)]
        if not self.v_t:
            self.v_t = [np.zeros_like(x) for x in delta_t]
        self.v_t = [self._second_moment(x, y) for x, y in zip(self.v_t, delta_t)]
        new_weights = [x + self.eta * y / (np.sqrt(z) + self.tau) for x, y, z in zip(self.current_weights, self.m_t, self.v_t)]
        self.current_weights = new_weights
        return fl.ndarrays_to_parameters(self.current_weights), metrics

    def server_step(self, agg: "DeviceWeights", counters_mean) -> "DeviceWeights":
        """The server optimizer on a FedAvg mean that already sits on the device -- ``aggregate_fit`` above, and the
        rank-per-GPU round driver, where the mean comes out of the weighted all-reduce and every rank takes the same step.
        `counters_mean`: the float64 weighted mean of the int64 counters (before truncation)."""
        cur = self.current.state
        if self.m_t is None:
            self.m_t, self.v_t = torch.zeros_like(cur), torch.zeros_like(cur)
            self.m_c = torch.zeros(self.cnt.shape, dtype=torch.float64, device=cur.device)
            self.v_c = torch.zeros_like(self.m_c)
        L.fedopt_step(self.MODE, cur, agg.state, self.m_t, self.v_t, self.eta, self.beta_1, self.beta_2, self.tau)
        d = counters_mean.double() - self.cnt.double()
        self.m_c = self.beta_1 * self.m_c + (1 - self.beta_1) * d
        self.v_c = self._second_moment(self.v_c, d)
        self.cnt = self.cnt.double() + self.eta * self.m_c / (torch.sqrt(self.v_c) + self.tau)
        self.current = DeviceWeights(cur, self.cnt.to(torch.int64))
        return DeviceWeights(cur.clone(), self.cnt.to(torch.int64))

    def __repr__(self):
        return f"{type(self).__name__}(accept_failures={self.accept_failures})"


def _mul(y):
    return y * y if torch.is_tensor(y) else np.multiply(y, y)


class FedAdagrad(FedOpt):
    MODE, DEFAULTS = 0, {"eta": 1e-1, "eta_l": 1e-1, "beta_1": 0.0, "beta_2": 0.0, "tau": 1e-9}

    def _second_moment(self, v, d):
        return v + _mul(d)


class FedAdam(FedOpt):
    MODE, DEFAULTS = 1, {"eta": 1e-1, "eta_l": 1e-1, "beta_1": 0.9, "beta_2": 0.99, "tau": 1e-9}

    def _second_moment(self, v, d):
        return self.beta_2 * v + (1 - self.beta_2) * _mul(d)


class FedYogi(FedOpt):
    MODE, DEFAULTS = 2, {"eta": 1e-2, "eta_l": 0.0316, "beta_1": 0.9, "beta_2": 0.99, "tau": 1e-3}

    def _second_moment(self, v, d):
        d2 = _mul(d)
        sign = torch.sign(v - d2) if torch.is_tensor(v) else np.sign(v - d2)
        return v - (1.0 - self.beta_2) * d2 * sign


def get_strategy(name, **kwargs):
    assert name in (CENTRALIZED_FL + PERSONALIZED_FL)
    cls = {"FedAvg": FedAvg, "FedICRA": FedICRA, "FedAdagrad": FedAdagrad, "FedAdam": FedAdam, "FedYogi": FedYogi}.get(name)
    if cls is None:
        raise NotImplementedError(name)
    return cls(**kwargs)


def fit_metrics_aggregation_fn(fit_metrics):
    return {k: v for _, client_metrics in fit_metrics for k, v in client_metrics.items()}


def get_evaluate_metrics_aggregation_fn(args, val_metrics):
    """flower_common.py:398-428: example-weighted and unweighted means of client_{i}_val_*."""
    def evaluate_metrics_aggregation_fn(evaluate_metrics):
        metrics = {k: v for _, cm in evaluate_metrics for k, v in cm.items()}
        weights = {}
        for client_id in range(args.min_num_clients):
            first = "client_{}_val_mean_{}".format(client_id, val_metrics[0])
            for n_ex, cm in evaluate_metrics:
                if first in cm:
                    weights["client_{}".format(client_id)] = n_ex

        def weighted_metric(name):
            tot = sum(weights.values())
            return sum(weights["client_{}".format(c)] * metrics["client_{}_{}".format(c, name)]
                       for c in range(args.min_num_clients)) / tot

        def mean_metric(name):
            return np.mean([metrics["client_{}_{}".format(c, name)] for c in range(args.min_num_clients)])

        metrics.update({"val_{}_{}".format(ci + 1, m): weighted_metric("val_{}_{}".format(ci + 1, m))
                        for ci in range(args.num_classes - 1) for m in val_metrics})
        metrics.update({"val_mean_{}".format(m): weighted_metric("val_mean_{}".format(m)) for m in val_metrics})
        metrics.update({"val_avg_mean_{}".format(m): mean_metric("val_mean_{}".format(m)) for m in val_metrics})
        return metrics
    return evaluate_metrics_aggregation_fn


# the reference defines the tree-energy losses in this module (flower_common.py:646-818); re-exported under the same names
from .tree_energy import MScaleAddTreeEnergyLoss, MScaleRecurveTreeEnergyLoss, TreeEnergyLoss, tv_loss  # noqa: E402,F401
