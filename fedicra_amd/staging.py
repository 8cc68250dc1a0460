"""Host -> device staging of training batches beside the compute stream.

The reference moves every batch with a synchronous ``.cuda()`` at the top of the iteration
(/root/reference/code/flower_pCE_2D.py:76-81, flower_common.py:568-573) behind a 4-worker DataLoader with pinned
memory (:303-304).  Here a batch that lives in (pinned) host memory is copied into a device staging pair on a side HIP
stream while the previous iteration computes; the iteration then takes it with one device copy into its static graph
inputs (12x3x512^2 fp32 = 37.7 MB: ~25 us on the device against ~0.7 ms over PCIe Gen5).  Batches that are already
device tensors (fedicra_amd.dataloaders.DeviceLoader, resident data) pass through untouched."""
from __future__ import annotations

import torch


import os

_NO_PREFETCH = os.environ.get("FEDICRA_NO_PREFETCH", "0") != "0"      # measurement switch: every copy serial, as in the reference


def _copy_stream(device):
    """The staging stream: the `staging` role of fedicra_amd/streams.py, which creates every role's stream in one place and in a fixed
    order.  (A host-to-device copy from pinned memory is a blit kernel on its stream's hardware queue: a staging stream that lands on
    the compute stream's queue is served in order WITH the compute kernels and the "hidden" 41 MB per step come back as 12-13 ms per
    round.  Round 5 tried to pick the stream by a one-off overlap trial and by HIP priority; the trial did not see the collision and
    the priority cost 10 % -- both are gone, the creation order is what is controlled now.)"""
    from . import streams
    dev = torch.device(device)
    if dev.type != "cuda":
        return None
    return streams.get("staging", dev)


class _Hip:
    """The four runtime objects the stager touches; tests/test_host_cpu.py swaps in host stand-ins to drive the ring logic
    (slot re-use, stale entries, out-of-order consumption) without a GPU."""
    Stream = staticmethod(_copy_stream)
    Event = staticmethod(lambda: torch.cuda.Event())
    current_stream = staticmethod(lambda: torch.cuda.current_stream())
    stream = staticmethod(lambda s: torch.cuda.stream(s))
    pinned = staticmethod(lambda t: t.is_pinned())


class BatchStager:
    """A RING of staging pairs per batch shape.  The side stream may overwrite a pair only after its last consumer has copied it
    out; making it wait for that with a stream-wait on an event of the compute stream costs most of the overlap on this runtime
    (tools/h2d_probe.py: 3.1-3.2 ms per step with the wait, 2.89 without, 3.69 with no prefetch at all, compute alone 2.79).  The
    host enqueues a whole round ahead of the GPU, so the ring is made longer than a round (12 pairs; 41 MB each at 12x3x512^2 -- HBM
    is 288 GB): the pair's release event is then long complete, which a host-side query sees, and the stream-wait is skipped.  If
    it is not complete yet the wait is made -- correctness never depends on the ring length."""

    SLOTS = 12

    def __init__(self, device, slots=None):
        self.device = torch.device(device)
        self.side = _Hip.Stream(self.device)
        self.slots = int(slots or self.SLOTS)
        self._bufs = {}            # (x shape, x dtype, y shape, y dtype) -> [[pair, ...], turn]; pair = [x_stage, y_stage, free event]
        # id(batch) -> (ready event, pair, batch): several copies may be in flight, each in its own pair.  The entry HOLDS the
        # batch: a host batch that was prefetched and then dropped by its loader must not have its id() re-used by a new
        # dict while the stale entry is alive (the new batch would be served the old pixels)
        self._pending = {}
        self.dropped_prefetches = 0
        self.pageable_ahead = False      # tests only: the round-3 behaviour (pageable batches copied ahead on the side stream too)
        self._held = None          # pair handed out by fetch() and not yet released
        self.h2d_bytes = 0

    @staticmethod
    def on_host(batch):
        return batch["image"].device.type == "cpu"

    def _pair(self, x, y):
        """The next staging pair of this batch shape's ring that nobody is waiting for.  A pair that still carries a
        prefetched batch (an entry of `_pending`) or that fetch() handed out and release() has not seen yet (`_held`) is
        SKIPPED; when every pair of a full ring is taken, the oldest pending entry is dropped -- its batch is then copied
        serially by fetch(), never served from a pair that another batch overwrote (a batch prefetched more
        than a ring length before its use used to come back with another batch's pixels)."""
        key = (tuple(x.shape), x.dtype, tuple(y.shape), y.dtype)
        ent = self._bufs.get(key)
        if ent is None:
            ent = self._bufs[key] = [[], 0]
        pairs, turn = ent
        busy = {id(p[1]) for p in self._pending.values()}
        if self._held is not None:
            busy.add(id(self._held))
        if len(pairs) < self.slots:                          # the ring fills up as it is used
            pairs.append([torch.empty(x.shape, dtype=x.dtype, device=self.device),
                          torch.empty(y.shape, dtype=y.dtype, device=self.device), None])
            return pairs[-1]
        for k in range(self.slots):
            cand = pairs[(turn + k) % self.slots]
            if id(cand) not in busy:
                ent[1] = (turn + k + 1) % self.slots
                return cand
        # every pair carries a batch somebody may still ask for: give up the OLDEST prefetch of this ring (dicts keep
        # insertion order), or -- a ring of one held pair -- grow past the nominal length rather than hand out live data
        for bid, (_, pr, _) in list(self._pending.items()):
            if any(pr is q for q in pairs):
                ready = self._pending.pop(bid)[0]
                pr[2] = ready                                # its copy on the side stream may still be in flight: the next
                self.dropped_prefetches += 1                 # writer of the pair waits for it (it waited for the reader before it)
                return pr
        pairs.append([torch.empty(x.shape, dtype=x.dtype, device=self.device),
                      torch.empty(y.shape, dtype=y.dtype, device=self.device), None])
        return pairs[-1]

    def prefetch(self, batch):
        """Start the copy of `batch` on the side stream; returns at once.  No-op for device-resident batches."""
        if batch is None or not self.on_host(batch) or _NO_PREFETCH or id(batch) in self._pending:
            return
        if len(self._pending) >= self.slots - 2:              # never more copies in flight than the ring can hold
            return
        x, y = batch["image"], batch["label"]
        if not (_Hip.pinned(x) and _Hip.pinned(y)) and not self.pageable_ahead:
            # pageable memory: the runtime bounces such a copy through its own pinned buffer and the "async" call blocks the
            # host -- nothing to overlap, and two streams bouncing at once is not something to lean on: fetch() copies it serially
            return
        pair = self._pair(x, y)
        if pair[2] is not None and not pair[2].query():
            self.side.wait_event(pair[2])             # the consumer that last read this pair has not run yet
        with _Hip.stream(self.side):
            pair[0].copy_(x, non_blocking=True)
            pair[1].copy_(y, non_blocking=True)
            ev = _Hip.Event()
            ev.record(self.side)
        self._pending[id(batch)] = (ev, pair, batch)
        self.h2d_bytes += x.numel() * x.element_size() + y.numel() * y.element_size()

    def fetch(self, batch):
        """-> (image, label) on the device for `batch`: a staging pair (call release() once it has been copied out), or the
        batch's own tensors when it is device-resident.  A batch that was not prefetched is copied on the current stream
        (the reference's serial behaviour)."""
        if not self.on_host(batch):
            return batch["image"], batch["label"]
        pend = self._pending.pop(id(batch), None)
        if pend is not None and pend[2] is not batch:        # cannot happen while the entry holds its batch; never serve it
            pend = None
        if pend is not None:
            _Hip.current_stream().wait_event(pend[0])
            self._held = pend[1]
            return pend[1][0], pend[1][1]
        x, y = batch["image"], batch["label"]
        pair = self._pair(x, y)
        if pair[2] is not None and not pair[2].query():
            _Hip.current_stream().wait_event(pair[2])   # (a consumer on another stream)
        # same stream as the pair's earlier consumers: ordered without an event
        pair[0].copy_(x, non_blocking=True)
        pair[1].copy_(y, non_blocking=True)
        self.h2d_bytes += x.numel() * x.element_size() + y.numel() * y.element_size()
        self._held = pair
        return pair[0], pair[1]

    def release(self):
        """The consumer has enqueued its copy out of the staging pair on the current stream."""
        if self._held is not None:
            ev = _Hip.Event()
            ev.record()
            self._held[2] = ev
            self._held = None
