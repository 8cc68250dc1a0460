"""Per-round parameter aggregation as ONE weighted all-reduce over xGMI (RCCL).

Replaces "N x gRPC upload of ~140 np.save blobs -> numpy weighted mean on the server -> N x gRPC
download" (/root/reference/code/flower_common.py:262 -> flwr FedAvg.aggregate_fit -> aggregate,
SURVEY.md 8-a16, 8e) with: every rank (= client = GPU) pre-multiplies its flat fp32 state by its
own n_k = len(trainloader), one ``all_reduce(SUM)`` over the flat buffer, one divide by sum(n_k).
The arithmetic is flwr's  reduce(add, [w_k * n_k]) / total  up to the summation order RCCL uses.
The int64 num_batches_tracked counters are reduced separately as int64 (n_k * counter), then
true-divided and truncated -- the reference's float64 round trip.

The collective runs on a dedicated side HIP stream, fenced with events against the training
stream, so the next round's batch staging (the reference's epoch pre-materialisation,
flower_pCE_2D.py:66-70) overlaps with it.  State is 7-10 MB: the ring is latency-bound, so one
large collective over the whole flat buffer (not one per tensor) is the right shape for xGMI.

``backend='gloo'`` (CPU tensors) is supported for the world_size>1 unit tests on a box without GPUs;
the device kernels are then replaced by the same torch expressions -- tests only, never the product.
"""
from __future__ import annotations

from typing import Optional

import torch

from . import streams
import torch.distributed as dist

from .flower_common import DeviceWeights


class WeightedAllReduce:
    def __init__(self, num_examples: int, device: Optional[torch.device] = None, group=None, constant_term=None,
                 timing: bool = False, always_collective: bool = False):
        """`constant_term` = (DeviceWeights, n): a fixed contribution n * state to the weighted sum, added by rank 0 --
        clients of the federation that no rank hosts (bench.py with fewer GPUs than clients).  `timing`: HIP events on the
        side stream around pre-scale / all-reduce / divide and on the training stream around the fence (bench.py's round
        split; `splits` collects one dict of event pairs per round).  `always_collective`: issue the two all-reduces even
        when the process group has ONE rank (a sum over one rank is the identity, so the result is bit-identical to the
        no-group path) -- the RCCL call path (communicator, side stream, event fence, its interplay with the captured
        training step) then executes on a single MI355X: tests/test_round5_gpu.py and bench.py's N = 1 line."""
        self.group = group
        self.timing = bool(timing)
        self.always_collective = bool(always_collective) and dist.is_initialized()
        self.splits = []
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.device = device
        self.on_gpu = device is not None and torch.device(device).type == "cuda"
        # one client per process (int) or several co-located clients (list): the rank contributes sum_k n_k * w_k
        self.n_local = [int(n) for n in num_examples] if isinstance(num_examples, (list, tuple)) else [int(num_examples)]
        self.n_k = sum(self.n_local)
        if self.world > 1:
            t = torch.tensor([self.n_k], dtype=torch.int64, device=device if self.on_gpu else "cpu")
            gathered = [torch.zeros_like(t) for _ in range(self.world)]
            dist.all_gather(gathered, t, group=group)        # n_k is constant: exchanged once
            self.all_n = [int(g.item()) for g in gathered]
        else:
            self.all_n = [self.n_k]
        self.total = sum(self.all_n)
        self.constant = None
        if constant_term is not None:
            cw, cn = constant_term
            self.total += int(cn)
            if self.rank == 0:
                self.constant = (cw, int(cn))
        self.side = streams.get("comm", device) if self.on_gpu else None         # (fedicra_amd/streams.py: every role's stream is made in one place)
        self._send = None
        self._cnt = None
        self._done = None
        self.counter_mean = None
        self.collectives_issued = 0                            # all-reduce calls handed to torch.distributed so far

    # ---------------------------------------------------------------------------------------------
    def start(self, weights):
        """Enqueue pre-scale + all-reduce + divide on the side stream; returns immediately.  `weights`: the
        DeviceWeights of this rank's client, or one per co-located client (same order as num_examples)."""
        many = list(weights) if isinstance(weights, (list, tuple)) else [weights]
        assert len(many) == len(self.n_local)
        weights = many[0]
        if self._send is None or self._send.shape != weights.state.shape:
            self._send = torch.empty_like(weights.state)
            self._cnt = torch.empty_like(weights.counters)
        if self.on_gpu:
            from . import _lib as L
            self.side.wait_stream(torch.cuda.current_stream())
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)] if self.timing else None
            with torch.cuda.stream(self.side):
                if ev:
                    ev[0].record(self.side)
                L.scale(weights.state, self._send, float(self.n_local[0]))
                torch.mul(weights.counters, self.n_local[0], out=self._cnt)
                for w, n in zip(many[1:], self.n_local[1:]):          # acc + w_k * n_k, left to right (numpy's order)
                    L.axpy(self._send, w.state, float(n))
                    self._cnt.add_(w.counters, alpha=n)
                if self.constant is not None:
                    L.axpy(self._send, self.constant[0].state, float(self.constant[1]))
                    self._cnt.add_(self.constant[0].counters, alpha=self.constant[1])
                if ev:
                    ev[1].record(self.side)
                if self.world > 1 or self.always_collective:
                    dist.all_reduce(self._send, op=dist.ReduceOp.SUM, group=self.group)
                    dist.all_reduce(self._cnt, op=dist.ReduceOp.SUM, group=self.group)
                    self.collectives_issued += 2
                if ev:
                    ev[2].record(self.side)
                L.scale(self._send, self._send, float(self.total), divide=True)
                if ev:
                    ev[3].record(self.side)
                    self.splits.append({"side": ev})
                self._done = torch.cuda.Event()
                self._done.record(self.side)
        else:   # gloo / CPU test path
            torch.mul(weights.state, float(self.n_local[0]), out=self._send)
            torch.mul(weights.counters, self.n_local[0], out=self._cnt)
            for w, n in zip(many[1:], self.n_local[1:]):
                self._send.add_(w.state * float(n))
                self._cnt.add_(w.counters, alpha=n)
            if self.constant is not None:
                self._send.add_(self.constant[0].state * float(self.constant[1]))
                self._cnt.add_(self.constant[0].counters, alpha=self.constant[1])
            if self.world > 1 or self.always_collective:
                dist.all_reduce(self._send, op=dist.ReduceOp.SUM, group=self.group)
                dist.all_reduce(self._cnt, op=dist.ReduceOp.SUM, group=self.group)
                self.collectives_issued += 2
            self._send.div_(float(self.total))

    def finish(self) -> DeviceWeights:
        """Fence the training stream behind the collective and return the global weights."""
        if self.on_gpu:
            if self.timing and self.splits:
                w0, w1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                w0.record()
                torch.cuda.current_stream().wait_event(self._done)
                w1.record()
                self.splits[-1]["fence"] = (w0, w1)
            else:
                torch.cuda.current_stream().wait_event(self._done)
        self.counter_mean = self._cnt.double() / self.total              # float64, as flwr's aggregate hands it back
        counters = self.counter_mean.to(torch.int64)                      # the clients' load truncates
        return DeviceWeights(self._send, counters)

    def split_ms(self):
        """Mean milliseconds per round of the recorded rounds (call after a device synchronize): pack = pre-scale (+ the
        co-located / absent clients' terms), collective = the two all-reduces (0 when one rank holds everything), divide, and
        `exposed` = how long the training stream actually stood at the fence; hidden = side-stream span - exposed."""
        if not self.splits:
            return None
        n = float(len(self.splits))
        pack = sum(s["side"][0].elapsed_time(s["side"][1]) for s in self.splits) / n
        coll = sum(s["side"][1].elapsed_time(s["side"][2]) for s in self.splits) / n
        div = sum(s["side"][2].elapsed_time(s["side"][3]) for s in self.splits) / n
        exposed = sum(s["fence"][0].elapsed_time(s["fence"][1]) for s in self.splits if "fence" in s) / n
        return {"pack": pack, "collective": coll, "divide": div, "exposed": exposed,
                "hidden": max(0.0, pack + coll + div - exposed)}

    def aggregate(self, weights: DeviceWeights) -> DeviceWeights:
        self.start(weights)
        return self.finish()


def init_process_group_from_env(backend: Optional[str] = None):
    """One process per GPU; RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the environment (torchrun)."""
    import os
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world <= 1:
        return 0, 0, 1
    rank, local = int(os.environ["RANK"]), int(os.environ.get("LOCAL_RANK", "0"))
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"     # 'nccl' IS RCCL on ROCm
    # test hooks: several ranks on ONE GPU (a 1-GPU box cannot host two RCCL ranks) exchange through gloo instead
    backend = os.environ.get("FEDICRA_DIST_BACKEND", backend)
    if "FEDICRA_FORCE_DEVICE" in os.environ:
        local = int(os.environ["FEDICRA_FORCE_DEVICE"])
    if torch.cuda.is_available():
        torch.cuda.set_device(local)
    if not dist.is_initialized():
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local, world
