"""The HIP streams of one client process: created in ONE place, in ONE fixed order.

A HIP stream lives on one of the GPU's hardware queues (GPU_MAX_HW_QUEUES, 4 by default, handed out round-robin as streams
are created), and work on two streams that share a queue is served in order, not side by side.  Through round 5 every component
made its own stream when it first needed one -- the batched LC forwards (flower_pCE_2D.MyClient), the batch stager, the
aggregation's side stream (comm.WeightedAllReduce), torch.cuda.graph's capture stream, the tree filter's branches -- so which
streams shared a queue depended on the order the components happened to be built in: the same bench measured 1 392 images/s with
a one-rank RCCL group alive and 1 298 without it (LOG.md (24)), and 66 against 72 ms of training per round between two
construction orders.  Here the first request for ANY role creates the streams of ALL roles of that device, in the order of
`ROLES`, distinct from one another and from the stream that is current at that moment; later requests return them.  Entry points
(bench.py, run_federated.py, the trainers' constructors) call `init()` before anything else makes a stream -- before the process
group, whose communicator brings streams of its own -- so that a process's roles sit at the same positions of torch's stream pool
in every run.  `describe()` puts the positions into the bench line.

The reference has one stream per process (the default one): /root/reference/code/flower_pCE_2D.py:76-81 copies, computes and
communicates in line."""
from __future__ import annotations

import threading

import torch

# creation order = position in the round-robin over the hardware queues, relative to the first one
ROLES = ("capture",      # torch.cuda.graph's capture stream: the training step itself when it is replayed
         "probe",        # the K-1 batched LC forwards beside the client's own forward
         "staging",      # host -> device batch copies beside the iteration
         "comm",         # the aggregation's collectives beside the next round's staging
         "tree0", "tree1", "tree2")   # parallel branches of the tree filter (the `_Ours` procedure)

_lock = threading.Lock()
_streams = {}             # device index -> {role: torch.cuda.Stream}


def init(device=None):
    """Create (once) the streams of every role on `device`; returns {role: stream}."""
    if not torch.cuda.is_available():
        return {}
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    if dev.type != "cuda":
        return {}
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    with _lock:
        got = _streams.get(idx)
        if got is not None:
            return got
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError("fedicra_amd.streams.init: called for the first time inside a stream capture -- the entry point must "
                               "call it before the first torch.cuda.graph()")
        seen = {torch.cuda.current_stream(idx).cuda_stream, torch.cuda.default_stream(idx).cuda_stream}
        got = {}
        for role in ROLES:
            for _ in range(64):                          # torch hands pool streams out round-robin: skip one that is already taken
                s = torch.cuda.Stream(device=idx)
                if s.cuda_stream not in seen:
                    break
            else:
                raise RuntimeError("fedicra_amd.streams: torch's stream pool returned a taken stream 64 times")
            seen.add(s.cuda_stream)
            got[role] = s
        assert len({s.cuda_stream for s in got.values()}) == len(ROLES)
        _streams[idx] = got
        return got


def get(role, device=None):
    """The stream of `role` (one of ROLES) on `device` (default: the current device)."""
    if role not in ROLES:
        raise KeyError(f"fedicra_amd.streams: unknown role {role!r} (roles: {ROLES})")
    return init(device)[role]


def describe(device=None):
    """{role: position in torch's stream pool} -- what the bench line carries, so that two runs can be told apart."""
    out = {}
    for role, s in init(device).items():
        sid = getattr(s, "stream_id", None)
        out[role] = None if sid is None else int((sid >> 5) & 31)      # c10 StreamId: index << 5 | type << 1 | 1
    return out
