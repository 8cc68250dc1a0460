#!/usr/bin/env python
"""bench.py -- images/s/client and ms/aggregation round of the FedICRA hot path on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W
(`python bench.py --gpus N` with N > 1 and no WORLD_SIZE in the environment launches those N ranks itself.)

Workload = BASELINE.json configs[2], the configuration the metric is quoted on: a federation of 8 FedICRA clients,
``unet_lc`` (UNet_LC(in, cls, pcs_num=1, emb_num=8, client_num=8, client_id=k), net_factory.py:24-26), 12 x C x 512 x 512
synthetic slices per batch (C = 3, 3 classes: the ODOC shape; ``--in-chns 1`` gives the FAZ shape), ONE client per GPU
(client k = rank k = GPU k, flower_runner.py:100-102).  Per-GPU work is fixed (weak scaling): with fewer than 8 GPUs the
clients that no rank hosts contribute their (constant) initial state to the weighted mean, so that the round a hosted
client sees -- 7 LC forwards per iteration, a global state that differs from its own -- is the 8-client round at every N.

A *step* is one local training iteration of every hosted client (flower_pCE_2D.py:51-181): zero-grad, forward, pCE,
the LC loss with its 7 no-grad forwards under the other clients' embeddings (:128-139), backward under the freeze
schedule (:84-101: head phase, then the last rep_iters = 3 iterations train the body), AdamW, poly LR -- on one batch
of 12 images already resident in HBM.  Every ``--round-iters`` (10, flower_runner.py:38-39) steps the round closes with
the *aggregation round*, all inside the timed region: pre-scaled flat state -> weighted all-reduce (RCCL over xGMI when
N > 1) on a side stream, overlapped with the next round's batch-list staging -> ``MyModel.set_weights`` = load of the
global state + FedICRA's adaptive local aggregation: one ALA epoch of forward / decoder-only backward / mixing update over
the client's ``--loader-batches`` training batches (flower_common.py:566-618).  ``config.ms_per_aggregation_round`` is
that interval (HIP events on the training stream).

Batches live in PINNED HOST memory (the reference's DataLoader, flower_pCE_2D.py:303-304) and cross PCIe inside the timed
region: fedicra_amd.staging.BatchStager copies batch i+1 on a side stream while iteration i computes (``--data resident``
keeps them in HBM instead; that rate is reported beside the headline as config.resident_images_per_sec).

value = images/s over ALL hosted clients (= images/s/client at N = 1; per-client figure in config), timed over the K
steps including their aggregation rounds, barrier + synchronize on both sides, max over ranks.  Rank 0 prints ONE JSON
line.  Extra objects on it (DESIGN.md "measurement"):
  roofline     -- the dominant kernel family priced against its roofline; launch durations measured live with HIP events
                  on the launch stream in an eager instrumented pass (no overhead subtraction; agrees with the committed
                  rocprofv3 summary of the same command under profiles/).
  cpu_baseline -- the CPU oracle (oracle/, kind "port") timed on this box's host cores on a bounded sample of the
                  same workload (rank 0, N == 1 only).
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: HBM3E 8 TB/s
MFMA_PEAK = {"bf16": 2500.0, "fp16": 2500.0, "fp32": 157.3}   # TFLOP/s dense (guide: 2.5 PF bf16/fp16, 157.3 TF fp32-input MFMA)
FEDERATION = 8                   # configs[2]: 8 clients


class _stdout_to_stderr:
    """RCCL prints a version banner on C stdout when its first communicator comes up (buffered: it used to surface at process
    exit, BEHIND the JSON line).  stdout carries exactly ONE line -- the result -- so file descriptor 1 points at stderr while a
    communicator is created, and the C streams are flushed before it is pointed back."""

    def __enter__(self):
        import ctypes
        self._libc = ctypes.CDLL(None)
        sys.stdout.flush()
        self._libc.fflush(None)
        self._saved = os.dup(1)
        os.dup2(2, 1)
        return self

    def __exit__(self, *exc):
        sys.stdout.flush()
        self._libc.fflush(None)
        os.dup2(self._saved, 1)
        os.close(self._saved)
        return False


def make_args(a, cid):
    return argparse.Namespace(strategy="FedICRA", amp=0, model="unet_lc", cid=cid, min_num_clients=FEDERATION,
                              num_classes=a.classes, img_class="faz" if a.in_chns == 1 else "odoc", base_lr=0.01,
                              max_iterations=30000, iters=a.round_iters, rep_iters=3, alpha=1.0, snapshot_path=None,
                              use_graph=not a.no_graph)


def make_loader(n_batches, a, cid, device, where):
    """The client's training batches: `host` = pinned host memory (every use crosses PCIe, staged on a side stream),
    `resident` = already in HBM."""
    from fedicra_amd.synth import phantom_batch
    out = []
    for i in range(n_batches):
        img, weak, _ = phantom_batch(a.batch, a.size, a.in_chns, a.classes, cid=cid, index=i)
        x, y = torch.from_numpy(img), torch.from_numpy(weak)
        out.append({"image": x.to(device), "label": y.to(device)} if where == "resident"
                   else {"image": x.pin_memory(), "label": y.pin_memory()})
    return out


class Federation:
    """The hosted client of this rank plus the round loop (fit -> aggregate -> set_weights) around it."""

    def __init__(self, a, rank, world, dev, dtype, data=None):
        from fedicra_amd.comm import WeightedAllReduce
        from fedicra_amd.flower_common import DeviceWeights, MyModel
        from fedicra_amd.flower_pCE_2D import MyClient
        from fedicra_amd.networks import net_factory
        from fedicra_amd.networks.unet import set_compute_dtype
        from fedicra_amd.synth import client_num_batches
        self.a, self.rank, self.world = a, rank, world
        cid = rank
        args = make_args(a, cid)
        torch.manual_seed(2022)                          # the reference seeds every client process with 2022
        net = net_factory(args, net_type="unet_lc", in_chns=a.in_chns, class_num=a.classes)
        set_compute_dtype(net, dtype)
        self.loader = make_loader(a.loader_batches, a, cid, dev, data or a.data)
        self.model = MyModel(args, net, self.loader, self.loader)
        # steady-state rounds: the one-off convergence loop of a client's FIRST personalised round (>= 11 ALA epochs,
        # flower_common.py:604-618) is not what "ms per aggregation round" means; every timed round runs exactly one epoch
        self.model.start_phase = False
        self.model.verbose = False                       # stdout carries the one JSON line
        self.client = MyClient(args, self.model, self.loader, self.loader)
        all_n = client_num_batches(FEDERATION, a.batch)          # FedAvg weights n_k = len(trainloader_k) of the 8 sites
        absent = None
        if world < FEDERATION:
            # clients 'world'..7 are hosted by nobody: their term of the weighted sum is n_k x (initial state), a constant
            n_abs = sum(all_n[world:])
            absent = (DeviceWeights(net.flat_state.clone(), net.flat_counters.clone()), n_abs)
        self.agg = WeightedAllReduce(all_n[cid], device=dev, constant_term=absent, timing=True,
                                     always_collective=bool(getattr(a, "rccl_single_rank", False)))
        self.backend = None
        if world > 1:
            # communicator creation (seconds, once) must not land inside a timed region, whatever --warmup is
            import torch.distributed as dist
            self.backend = dist.get_backend()
            if not os.environ.get("FEDICRA_DIST_BACKEND"):          # (test hook: several ranks on one GPU through gloo)
                assert self.backend == "nccl", f"multi-GPU bench must exchange over RCCL, got backend {self.backend!r}"
            with _stdout_to_stderr():
                dist.all_reduce(torch.zeros(1, device=dev if self.backend == "nccl" else "cpu"))
                torch.cuda.synchronize()
        self.iter_global = 60                            # > 50: the ALA branch runs (flower_common.py:524-526)
        self.agg_events = []
        self.train_events = []
        self._loaded = None
        self.model.__dict__["_timing_mark"] = self._mark

    def _mark(self, what):
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        self._loaded = e

    def run_steps(self, nsteps):
        a, c = self.a, self.client
        done = 0
        while done < nsteps:
            it = min(a.round_iters, nsteps - done)
            c.args.iters = it
            t0 = torch.cuda.Event(enable_timing=True)
            t0.record()
            c._train({"iter_global": self.iter_global, "iters": it, "eval_iters": 10 * it, "batch_size": a.batch,
                      "stage": "fit"})
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            self.agg.start(self.model.get_device_weights())      # side stream: pre-scale, all-reduce, divide
            c.sampled_batches = list(c.trainloader)              # overlapped: next round's batch staging (epoch list)
            self.model.stage_ahead(self.loader[0])               # ... and the first batch of the ALA epoch on its way over PCIe
            glob = self.agg.finish()
            self._loaded = None
            self.model.set_weights(glob, {"iter_global": self.iter_global})    # global load + ALA epoch
            e1.record()
            self.agg_events.append((e0, e1, self._loaded))
            self.train_events.append((t0, e0, it))
            self.iter_global += 1
            done += it

    def timed(self, warmup, steps, dist, windows=1):
        """W untimed warm-up steps, then `windows` back-to-back windows of EXACTLY `steps` steps, each bracketed by barrier +
        synchronize on both sides and reduced with MAX over ranks.  -> (elapsed of the MEDIAN window, its mean aggregation
        round in ms, [elapsed of every window]).  The event lists round_split() reads are those of the median window.
        The warm-up is rounded UP to whole rounds and to at least two of them (self.warmup_run): a captured step exists from
        the third use of its freeze phase on and the ALA epoch's graph from the second round on, so a 5-step warm-up left
        three captures inside the first timed window (round 4: 1 187 against 1 386 images/s)."""
        ri = self.a.round_iters
        self.warmup_run = max(-(-int(warmup) // ri) * ri, 2 * ri)
        self.run_steps(self.warmup_run)
        res = []
        for _ in range(max(1, windows)):
            torch.cuda.synchronize()
            if self.world > 1:
                dist.barrier()
            self.agg_events, self.train_events, self.agg.splits = [], [], []
            t0 = time.perf_counter()
            self.run_steps(steps)
            torch.cuda.synchronize()
            if self.world > 1:
                dist.barrier()
            elapsed = time.perf_counter() - t0
            if self.world > 1:
                t = torch.tensor([elapsed], dtype=torch.float64, device=self.model.model.flat_state.device)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                elapsed = float(t.item())
            res.append((elapsed, self.agg_events, self.train_events, self.agg.splits))
        order = sorted(range(len(res)), key=lambda i: res[i][0])
        elapsed, self.agg_events, self.train_events, self.agg.splits = res[order[(len(order) - 1) // 2]]
        agg_ms = [e0.elapsed_time(e1) for e0, e1, _ in self.agg_events]
        return elapsed, sum(agg_ms) / max(len(agg_ms), 1), [r[0] for r in res]

    def round_split(self):
        """Mean ms per round of the last timed() call: local training (round_iters steps), then the aggregation round cut
        into pack (pre-scale on the side stream) / collective (all-reduce) / unpack (divide + the load of the global state
        into the model, up to the event after it) / ala (the rest of set_weights: one ALA epoch)."""
        n = float(max(len(self.agg_events), 1))
        sp = self.agg.split_ms() or {"pack": 0.0, "collective": 0.0, "divide": 0.0, "exposed": 0.0, "hidden": 0.0}
        train = sum(t0.elapsed_time(t1) for t0, t1, _ in self.train_events) / n
        steps = sum(it for _, _, it in self.train_events) / n
        to_loaded = sum(e0.elapsed_time(l) for e0, _, l in self.agg_events if l is not None) / n
        total = sum(e0.elapsed_time(e1) for e0, e1, _ in self.agg_events) / n
        out = {"train": train, "train_steps_per_round": steps, "pack": sp["pack"], "collective": sp["collective"],
               "unpack": max(0.0, to_loaded - sp["pack"] - sp["collective"]), "ala": total - to_loaded,
               "aggregation_total": total, "side_stream_exposed": sp["exposed"], "overlap_hidden": sp["hidden"]}
        return {k: round(v, 3) for k, v in out.items()}


# ---------------------------------------------------------------------------------------------------------------------
# --workload c4: BASELINE.json configs[3] -- 4 clients, 3D U-Net 128^3 CT patches, bf16, one client per MI355X
# ---------------------------------------------------------------------------------------------------------------------
C4_CLIENTS = 4
C5_CLIENTS = 8
C4_FWD_GF = 289.14          # conv FLOPs of one forward of unet_3D(1,2) at 128^3, per volume


def client_sizes_3d(k):
    """n_k of the k sites: the FAZ site sizes / batch the 2D federation uses, cycled."""
    base = [21, 13, 17, 59, 3]
    return [base[i % len(base)] for i in range(k)]


class Volumes:
    """The hosted 3D client: ``unet_3D(n_classes=2, in_channels=1)`` (networks/unet_3D.py:20-94) on 2 x 1 x S^3 patches.  The
    reference never trains its 3D networks in a federation (nothing calls net_factory_3d), so the step is
    the 2D client loop carried over: zero-grad, forward, CE, backward, SGD(momentum 0.9, wd 1e-4 -- the reference's 3D trainers'
    optimizer) with the poly LR, and FedAvg every ``round_iters`` steps (pre-scaled flat state -> weighted all-reduce -> load)."""

    def __init__(self, a, rank, world, dev, dtype, kind="c4"):
        from fedicra_amd.comm import WeightedAllReduce
        from fedicra_amd.flower_common import DeviceWeights
        from fedicra_amd.networks.net_factory_3d import net_factory_3d
        from fedicra_amd.networks.unet import set_compute_dtype
        from fedicra_amd.optim import FusedAdamW, FusedSGD
        self.a, self.rank, self.world, self.dev, self.kind, self.dtype_name = a, rank, world, dev, kind, dtype
        self.federation = C4_CLIENTS if kind == "c4" else C5_CLIENTS
        torch.manual_seed(2022)
        self.scaler = None
        if kind == "c4":
            self.net = net_factory_3d("unet_3D", 1, 2).to(dev).train()
            set_compute_dtype(self.net, dtype)
        else:
            # configs[4]: "3D U-Net + per-client adapter heads, fp16": unet_3D_lc (channel selection on the deepest block +
            # an auxiliary Conv3d head, networks/unet_3D.py) in the reference's autocast dtype with its GradScaler
            # (flower_pCE_2D.py:47-48,143-146), AdamW like the 2D clients (:55)
            from fedicra_amd.amp import GradScaler
            from fedicra_amd.networks.unet_3D import unet_3D_lc
            self.net = unet_3D_lc(n_classes=2, in_channels=1, client_num=C5_CLIENTS, client_id=rank).to(dev)
            self.net.set_compute_dtype(dtype).train()
            self.scaler = GradScaler()
        g = torch.Generator().manual_seed(2022 + 1000 * rank)
        S = a.size
        self.batches = [(torch.rand(a.batch, 1, S, S, S, generator=g).to(dev),
                         (torch.rand(a.batch, S, S, S, generator=g) > 0.5).to(torch.uint8).to(dev)) for _ in range(2)]
        if kind == "c4":
            self.opt = FusedSGD(self.net, lr=0.01, base_lr=0.01, max_iterations=30000)
        else:
            self.opt = FusedAdamW(self.net, lr=0.01, base_lr=0.01, max_iterations=30000)
        n_all = client_sizes_3d(self.federation)
        absent = None
        if world < self.federation:
            absent = (DeviceWeights(self.net.flat_state.clone(), self.net.flat_counters.clone()), sum(n_all[world:]))
        self.agg = WeightedAllReduce(n_all[rank], device=dev, constant_term=absent, timing=True)
        self.backend = None
        if world > 1:
            import torch.distributed as dist
            self.backend = dist.get_backend()
            if not os.environ.get("FEDICRA_DIST_BACKEND"):
                assert self.backend == "nccl", f"multi-GPU bench must exchange over RCCL, got backend {self.backend!r}"
            with _stdout_to_stderr():
                dist.all_reduce(torch.zeros(1, device=dev if self.backend == "nccl" else "cpu"))
                torch.cuda.synchronize()
        self.it = 0
        self.agg_events = []
        self.graphs, self.graph_error = {}, None

    def _step_eager(self, key):
        from fedicra_amd import ops
        x, y = self.batches[key]
        ops.begin_iteration(self.dev)
        self.opt.zero_grad()
        out = self.net(x)                                        # NCDHW view of fp32 NDHWC logits
        if isinstance(out, (list, tuple)):                       # unet_3D_lc returns UNet_LC's list; the step reads [0]
            out = out[0]
        lg = out.permute(0, 2, 3, 4, 1)
        N, D, H, W, Cc = lg.shape
        loss = ops.ce_loss(lg.reshape(N * D, H, W, Cc), y.reshape(N * D, H, W), 255)
        if self.scaler is not None:                              # fp16: scale -> backward -> unscale / inf check -> step -> update
            self.scaler.scale(loss).backward()
            self.scaler.step(self.opt)
            self.scaler.update()
        else:
            loss.backward()
            self.opt.step()
        self.opt.advance_lr()

    def step(self):
        """One iteration; with --no-graph off, captured per batch buffer (the two resident batches alternate) after one eager
        pass and replayed from then on.  A capture that fails leaves the iteration eager and says so in the bench line."""
        from fedicra_amd import ops
        key = self.it % 2
        st = None if self.a.no_graph else self.graphs.get(key, "new")
        if st is None or st == "eager":
            self._step_eager(key)
        elif st == "new":
            self._step_eager(key)
            self.graphs[key] = "warm"
        elif st == "warm":
            torch.cuda.synchronize()
            ops.reserve_graph_tables()
            ops.bump_weights_epoch()
            g = torch.cuda.CUDAGraph()
            try:
                with torch.cuda.graph(g, stream=_capture_stream(), capture_error_mode="thread_local"):
                    self._step_eager(key)
                self.graphs[key] = g
                g.replay()
                ops.bump_weights_epoch()
            except Exception as e:                               # noqa: BLE001 -- reported, not hidden
                self.graphs[key] = "eager"
                self.graph_error = repr(e)[:300]
                torch.cuda.synchronize()
                self._step_eager(key)
        else:
            st.replay()
            ops.bump_weights_epoch()
        self.it += 1

    def run_steps(self, nsteps):
        from fedicra_amd.flower_common import DeviceWeights
        a, done = self.a, 0
        while done < nsteps:
            it = min(a.round_iters, nsteps - done)
            for _ in range(it):
                self.step()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            glob = self.agg.aggregate(DeviceWeights(self.net.flat_state, self.net.flat_counters))
            self.net.flat_state.copy_(glob.state)
            self.net.flat_counters.copy_(glob.counters)
            e1.record()
            self.agg_events.append((e0, e1))
            done += it

    def timed(self, warmup, steps, dist):
        self.run_steps(warmup)
        torch.cuda.synchronize()
        if self.world > 1:
            dist.barrier()
        self.agg_events, self.agg.splits = [], []
        t0 = time.perf_counter()
        self.run_steps(steps)
        torch.cuda.synchronize()
        if self.world > 1:
            dist.barrier()
        elapsed = time.perf_counter() - t0
        if self.world > 1:
            t = torch.tensor([elapsed], dtype=torch.float64, device=self.dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
        agg_ms = [e0.elapsed_time(e1) for e0, e1 in self.agg_events]
        return elapsed, sum(agg_ms) / max(len(agg_ms), 1)

    def roofline(self, dtype_name):
        from fedicra_amd import _lib as L
        self.a.no_graph = True                                   # instrumented launches are eager
        self.step()
        L.profile_begin(subtract_overhead=False)
        for _ in range(2):
            self.step()
        prof = L.profile_end().summary()
        pk_f, pk_b = MFMA_PEAK[dtype_name] * 1e12, HBM_PEAK_GBS * 1e9
        fam = {}
        for k, v in prof.items():
            f = fam.setdefault(k[0], {"calls": 0, "ms": 0.0, "flops": 0.0, "bytes": 0.0, "ideal": 0.0})
            for q in ("calls", "ms", "flops", "bytes"):
                f[q] += v[q]
            f["ideal"] += max(v["flops"] / pk_f, v["bytes"] / pk_b) * 1e3
        total_ms = sum(v["ms"] for v in fam.values())
        name, dom = max(fam.items(), key=lambda kv: kv[1]["ms"])
        avg_ms = dom["ms"] / dom["calls"]
        flops, nbytes = dom["flops"] / dom["calls"], dom["bytes"] / dom["calls"]
        ai = flops / max(nbytes, 1.0)
        if ai >= pk_f / pk_b:
            bound, ach, peak, unit = "mfma", flops / (avg_ms * 1e-3) / 1e12, MFMA_PEAK[dtype_name], "TFLOP/s"
        else:
            bound, ach, peak, unit = "hbm", nbytes / (avg_ms * 1e-3) / 1e9, HBM_PEAK_GBS, "GB/s"
        table = os.environ.get("FEDICRA_BENCH_TABLE")
        if table:
            with open(table + getattr(self.a, "table_suffix", ""), "w") as f:
                f.write(f"# per-launch-shape roofline of one {self.kind} training iteration (2 instrumented iterations, HIP events, {dtype_name})\n")
                for k, v in sorted(prof.items(), key=lambda kv: -kv[1]["ms"]):
                    us = v["ms"] / v["calls"] * 1e3
                    idl = max(v["flops"] / pk_f, v["bytes"] / pk_b) / v["calls"] * 1e6
                    f.write(f"{'/'.join(map(str, k)):64s} {v['calls']:4d} {us:9.1f} us  ideal {idl:8.1f}  frac {idl / max(us, 1e-9):6.3f}  "
                            f"{v['flops'] / v['calls'] / (us * 1e-6) / 1e12:8.1f} TF/s {v['bytes'] / v['calls'] / (us * 1e-6) / 1e9:8.1f} GB/s\n")
        return {"bound": bound, "achieved": round(ach, 2), "peak": peak, "unit": unit, "frac": round(ach / peak, 4), "traffic": None,
                "kernel": f"{name} (all shapes of the unet_3D iteration)/{dtype_name}", "avg_us": round(avg_ms * 1e3, 2),
                "launches_per_step": dom["calls"] / 2.0, "arithmetic_intensity_flop_per_byte": round(ai, 1),
                "algorithmic_flops_per_launch": flops, "algorithmic_bytes_per_launch": nbytes,
                "share_of_timed_launch_time": round(dom["ms"] / total_ms, 4),
                "min_roofline_frac": round(dom["ideal"] / max(dom["ms"], 1e-9), 4),
                "min_roofline_frac_all_conv": round(sum(v["ideal"] for k, v in fam.items() if k.startswith("conv")) /
                                                    max(sum(v["ms"] for k, v in fam.items() if k.startswith("conv")), 1e-9), 4),
                "conv_flops_per_step": sum(v["flops"] for k, v in fam.items() if k.startswith("conv")) / 2.0,
                "kernel_time_breakdown_ms_per_step": {k: round(v["ms"] / 2.0, 3) for k, v in sorted(fam.items())},
                "profile_command": f"python bench.py --workload {self.kind} --roofline-only  (profiles/*_{self.kind}_*)"}


def cpu_baseline_c4(a):
    """oracle.unet3d_ref.RefUNet3D (kind "port") on the host cores: one warm-up + one timed iteration (forward, CE, backward, SGD) on
    ONE volume of half the edge (1 x 64^3 = 1/16 of the timed batch's voxels), scaled linearly in the voxel count."""
    from oracle.unet3d_ref import RefUNet3D
    cores = min(os.cpu_count() or 1, int(os.environ.get("FEDICRA_CPU_THREADS", "32")))
    torch.set_num_threads(cores)
    torch.manual_seed(2022)
    S = max(16, a.size // 2)
    m = RefUNet3D(n_classes=2, in_channels=1).train()
    opt = torch.optim.SGD(m.parameters(), lr=0.01, momentum=0.9, weight_decay=1e-4)
    x, y = torch.rand(1, 1, S, S, S), (torch.rand(1, S, S, S) > 0.5).long()
    ts = []
    for _ in range(2):
        t0 = time.perf_counter()
        opt.zero_grad()
        loss = torch.nn.functional.cross_entropy(m(x), y)
        loss.backward()
        opt.step()
        ts.append(time.perf_counter() - t0)
    vox_ratio = float(a.size ** 3) / float(S ** 3)
    return {"value": round(1.0 / (ts[-1] * vox_ratio), 4), "unit": "volumes/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"oracle.unet3d_ref.RefUNet3D (torch {torch.__version__} CPU fp32, {cores} threads): 1 warm-up + 1 timed training "
                      f"iteration on 1x1x{S}^3, scaled by the voxel ratio {vox_ratio:.0f} to a {a.size}^3 volume"}


def volumes_line(a, vol, elapsed, agg_ms, world, roof):
    """The JSON line of a 3D workload (configs[3] / configs[4]); `roof` = Volumes.roofline() or None."""
    kind, fed_n = vol.kind, vol.federation
    value = a.steps * a.batch * world / elapsed
    step_s = (elapsed - agg_ms * 1e-3 * len(vol.agg_events)) / a.steps               # training part of a step
    f_train = 3.0 * C4_FWD_GF * (a.size / 128.0) ** 3 * a.batch                      # GF per step (dgrad + wgrad for every conv)
    if roof is not None and roof.get("conv_flops_per_step"):
        f_train = roof["conv_flops_per_step"] / 1e9                                  # the launches' own algorithmic count
    captured = any(not isinstance(g, str) for g in vol.graphs.values())
    model = "unet_3D(n_classes=2, in_channels=1)" if kind == "c4" else "unet_3D_lc (unet_3D + channel selection + adapter head), fp16 + GradScaler"
    optim = "SGD momentum 0.9" if kind == "c4" else "GradScaler + AdamW"
    line = {"metric": f"volumes/sec/client (3D U-Net 128^3 local training, configs[{3 if kind == 'c4' else 4}]) ; ms/aggregation round in config",
            "value": round(value, 3), "unit": "volumes/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(elapsed / a.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": {"bf16": "bf16", "fp16": "f16", "fp32": "f32"}[vol.dtype_name], "data": "synthetic",
            "config": {"workload": f"BASELINE.json configs[{3 if kind == 'c4' else 4}]: {fed_n} clients FedAvg, {model}, "
                                   f"{a.batch}x1x{a.size}^3 patches per batch, one client per MI355X ({world} hosted); step = one "
                                   f"local iteration (fwd, CE, bwd, {optim}, poly LR), "
                                   f"{'one hipGraph per resident batch buffer' if captured else 'eager launches'}; round = "
                                   f"{a.round_iters} steps + weighted all-reduce + load, all timed; data resident in HBM",
                       "clients_hosted": world, "federation": fed_n, "global_batch": a.batch * world,
                       "volumes_per_sec_per_client": round(value / world, 3), "ms_per_aggregation_round": round(agg_ms, 3),
                       "conv_tflops_per_gpu": round(f_train / step_s / 1e3, 2),
                       "frac_of_mfma_peak": round(f_train / step_s / 1e3 / MFMA_PEAK[vol.dtype_name], 4),
                       "frac_of_mfma_peak_is": "algorithmic conv FLOPs of a step / time of the training steps (aggregation excluded)",
                       "hipgraph": captured, "hipgraph_error": vol.graph_error, "parallelism": f"fed-dp{world}"}}
    if vol.scaler is not None:
        line["config"]["grad_scale_after"] = float(vol.scaler.get_scale())
    if world > 1:
        sp = vol.agg.split_ms() or {}
        line["config"].update({"rccl_ranks": world if vol.backend == "nccl" else 0, "dist_backend": vol.backend,
                               "allreduce_us_per_round": round(sp.get("collective", 0.0) * 1e3, 1)})
    if roof is not None:
        line["roofline"] = roof
    return line


def _capture_stream():
    from fedicra_amd import streams
    return streams.get("capture")


def _stream_positions():
    """{role: position in torch's stream pool} of this process's streams (fedicra_amd/streams.py: one place, one fixed order)."""
    from fedicra_amd import streams
    return streams.describe()


def volumes_leg(a, rank, world, dev, dist, kind, steps=20, warmup=12):
    """configs[3] / configs[4] beside the headline: a short run of the 3D client OUTSIDE the c3 timed
    region -- `warmup` untimed steps (both resident batch buffers captured), `steps` timed ones incl. their FedAvg rounds, then
    the instrumented eager iterations of its roofline -- condensed to one object for config.c4 / config.c5."""
    import copy
    b = copy.copy(a)
    b.size, b.batch, b.steps, b.warmup = 128, 2, steps, warmup
    b.table_suffix = "." + kind             # FEDICRA_BENCH_TABLE: the legs write <path>.c4 / <path>.c5 beside the headline's own table
    dtype = "bf16" if kind == "c4" else "fp16"
    t0 = time.perf_counter()
    vol = Volumes(b, rank, world, dev, dtype, kind=kind)
    elapsed, agg_ms = vol.timed(b.warmup, b.steps, dist)
    try:
        roof = vol.roofline(dtype)
    except Exception as e:  # noqa: BLE001
        roof = {"error": repr(e)[:300]}
    line = volumes_line(b, vol, elapsed, agg_ms, world, roof if "error" not in roof else None)
    c = line["config"]
    out = {"workload": c["workload"], "volumes_per_sec": line["value"], "ms_per_step": line["ms_per_step"], "dtype": line["dtype"],
           "steps": b.steps, "warmup": b.warmup, "ms_per_aggregation_round": c["ms_per_aggregation_round"],
           "conv_tflops_per_gpu": c["conv_tflops_per_gpu"], "frac_of_mfma_peak": c["frac_of_mfma_peak"],
           "hipgraph": c["hipgraph"], "hipgraph_error": c["hipgraph_error"]}
    if "grad_scale_after" in c:
        out["grad_scale_after"] = c["grad_scale_after"]
    if "error" in roof:
        out["roofline_error"] = roof["error"]
    else:
        out.update({"min_roofline_frac": roof["min_roofline_frac"], "min_roofline_frac_all_conv": roof["min_roofline_frac_all_conv"],
                    "dominant_kernel": roof["kernel"], "dominant_frac": roof["frac"], "dominant_bound": roof["bound"],
                    "kernel_time_breakdown_ms_per_step": roof["kernel_time_breakdown_ms_per_step"]})
    out["seconds"] = round(time.perf_counter() - t0, 1)
    del vol
    torch.cuda.empty_cache()
    return out


def main_c4(a, rank, local, world, dev, dist):
    kind = a.workload
    dtype = a.dtype if kind == "c4" else ("fp16" if a.dtype == "bf16" else a.dtype)     # configs[4] names fp16
    assert world <= (C4_CLIENTS if kind == "c4" else C5_CLIENTS), "one client per GPU"
    vol = Volumes(a, rank, world, dev, dtype, kind=kind)
    if a.roofline_only:
        roof = vol.roofline(dtype)
        if rank == 0:
            print(json.dumps({"roofline": roof}), flush=True)
        return
    elapsed, agg_ms = vol.timed(a.warmup, a.steps, dist)
    if rank == 0:
        roof = None
        if not a.no_roofline:
            try:
                roof = vol.roofline(dtype)
            except Exception as e:  # noqa: BLE001
                roof = {"error": repr(e)}
        line = volumes_line(a, vol, elapsed, agg_ms, world, roof)
        if world == 1 and not a.no_cpu_baseline and kind == "c4":
            line["cpu_baseline"] = cpu_baseline_c4(a)
        print(json.dumps(line), flush=True)


def cpu_baseline(a):
    """The CPU oracle (oracle/, kind "port") on this box's host cores, on a bounded sample of the same workload:
    the FedICRA local-training iteration (LC forwards included) in the timed run's head : body mix, then one aggregation
    round (numpy FedAvg over 8 client states + one ALA epoch), on the FULL 12-image batch (rounds 3-5 extrapolated
    from a third of it): a warm-up, one head-phase and one body-phase iteration are about a minute on the GPU box's host; the
    thread-scaling table that picks the thread count runs on 4 images (FEDICRA_CPU_BATCH overrides the sample's batch)."""
    import numpy as np
    from oracle import fed_ref
    from oracle.unet_ref import RefUNetLC
    from fedicra_amd.synth import client_num_batches, phantom_batch
    # torch's CPU conv path stops scaling (and collapses from oversubscription) far below the 256 hardware
    # threads of the GPU box's host: use at most 32 threads and report that number as `cores`.
    torch.manual_seed(2022)
    B = max(1, min(a.batch, int(os.environ.get("FEDICRA_CPU_BATCH", str(a.batch)))))
    BS = min(B, 4)                                                             # images of the thread-scaling table
    m = RefUNetLC(a.in_chns, a.classes, 1, FEDERATION, FEDERATION, 0, heads=1)
    batches = []
    for i in range(2):
        img, weak, _ = phantom_batch(B, a.size, a.in_chns, a.classes, cid=0, index=i)
        batches.append({"image": torch.from_numpy(img), "label": torch.from_numpy(weak)})
    # thread-scaling table: one train-mode forward + backward of the same model on the same B images per thread
    # count, after a warm-up at that count; the sample below then runs at the FASTEST count, which is what `cores` reports
    ncpu = os.cpu_count() or 1
    forced = os.environ.get("FEDICRA_CPU_THREADS")
    scaling = {}
    if forced:
        cores = min(ncpu, int(forced))
    else:
        xb = (batches[0]["image"] if a.in_chns != 1 else batches[0]["image"].unsqueeze(1))[:BS]
        for th in sorted({t for t in (8, 16, 32, 64, 128, ncpu // 2, ncpu) if 0 < t <= ncpu}):
            torch.set_num_threads(th)
            ts = []
            for _ in range(2):
                t0 = time.perf_counter()
                m.zero_grad()
                m(xb)[0].square().mean().backward()
                ts.append(time.perf_counter() - t0)
            scaling[th] = round(BS / ts[-1], 3)
        m.zero_grad()
        cores = max(scaling, key=scaling.get)
    torch.set_num_threads(cores)
    st = fed_ref.TrainState(0.01)
    kw = dict(num_classes=a.classes, base_lr=0.01, max_iterations=30000, img_class="faz" if a.in_chns == 1 else "odoc",
              strategy="FedICRA", alpha=1.0, cid=0, num_clients=FEDERATION)
    fed_ref.local_train(m, st, batches, iters=1, rep_iters=0, **kw)          # warm-up: thread pool, allocator, primitives
    t_head, t_body = [], []
    for rep in (0, 1):                                                         # one head-phase + one body-phase iteration
        t0 = time.perf_counter()
        fed_ref.local_train(m, st, batches, iters=1, rep_iters=rep, **kw)
        (t_body if rep else t_head).append(time.perf_counter() - t0)
    head_frac = (a.round_iters - 3) / float(a.round_iters)
    s_iter = head_frac * float(np.median(t_head)) + (1.0 - head_frac) * float(np.median(t_body))
    # aggregation round: flwr's aggregate restated in numpy over the 8 clients' wire payloads + one ALA epoch
    w = fed_ref.get_weights(m)
    n_k = client_num_batches(FEDERATION, a.batch)
    t0 = time.perf_counter()
    glob = fed_ref.fedavg_aggregate([([x + np.float32(0.001 * k) if x.dtype == np.float32 else x for x in w], n_k[k])
                                     for k in range(FEDERATION)])
    t_agg = time.perf_counter() - t0
    t0 = time.perf_counter()
    fed_ref.set_weights_ala(m, glob, batches[:1], num_classes=a.classes, iter_global=60, start_phase=False,
                            img_class="faz" if a.in_chns == 1 else "odoc")
    t_ala_batch = (time.perf_counter() - t0) * (a.batch / float(B))           # one ALA batch of the full size
    return {"value": round(B / s_iter, 3), "unit": "images/s", "cores": torch.get_num_threads(), "kind": "port",
            "ms_per_aggregation_round": round((t_agg + t_ala_batch * a.loader_batches) * 1e3, 1),
            "ms_fedavg_numpy_k8": round(t_agg * 1e3, 2), "host_hardware_threads": ncpu,
            "thread_scaling_fwd_bwd_images_per_sec": scaling or None,
            "sec_per_head_iteration": round(float(np.median(t_head)), 3), "sec_per_body_iteration": round(float(np.median(t_body)), 3),
            "sample": f"oracle.fed_ref.local_train on RefUNetLC (torch {torch.__version__} CPU fp32, {cores} threads = the fastest of the "
                      f"measured thread-scaling table, taken on {BS} images): 1 warm-up + 1 head-phase + 1 body-phase FedICRA iteration of {B}x{a.in_chns}x{a.size}x{a.size} (the whole batch) incl. "
                      f"the 7 LC forwards, weighted {a.round_iters - 3}:3 like the timed rounds; aggregation = numpy FedAvg K=8 + one "
                      f"ALA batch of {B} images scaled to {a.loader_batches} batches of {a.batch}"}


def dice_leg(with_cpu=True, rounds=12):
    """Metric leg 3, "Dice vs CPU ref" (/root/reference/code/val_2D.py:9-74, flower_common.py:
    122-136): OUTSIDE the timed region, the miniature federation of fedicra_amd/minifed.py -- 2 FedAvg clients x `rounds`
    rounds x 8 local iterations on 4x1x64x64 phantoms (BASELINE configs[0]'s shape), weighted aggregation, `evaluate` of
    client 0 on 16 dense-mask cases -- trained THREE times from the same seeded state with the same dropout masks: on the HIP
    path in fp32 (parity mode) and in bf16 (the timed mode), and on the CPU port of the reference (oracle/minifed_ref.py;
    `with_cpu`, part of the cpu_baseline leg).  After ~100 sign-like AdamW steps per client the reference's own Dice moves
    with the dropout seed and the host's thread count (tests/golden/g19_minifed_dice.npz: the reference's own classes over
    8 seeds x {8, 1} threads); that spread is printed beside the deltas."""
    import numpy as np
    from fedicra_amd.minifed import make_data, run_hip
    from oracle.unet_ref import seeded_state      # the seeded initial state is DATA for both sides (weights are not stored)
    data, val = make_data()
    init = lambda net: seeded_state(net, 2022)
    t0 = time.perf_counter()
    out = {"rounds": rounds, "clients": 2, "local_iterations": 8, "shape": "4x1x64x64, 16 validation cases",
           "masks": "identical (host generator, seed 100*round + cid)"}
    h32 = run_hip(data, val, dtype="fp32", rounds=rounds, init_state=init, max_iterations=400)
    h16 = run_hip(data, val, dtype="bf16", rounds=rounds, init_state=init, max_iterations=400)
    out["hip_fp32"], out["hip_bf16"] = round(h32["dice"], 6), round(h16["dice"], 6)
    out["abs_delta_bf16_vs_fp32"] = round(abs(h16["dice"] - h32["dice"]), 6)
    if with_cpu:
        from oracle.minifed_ref import run_oracle
        cores = min(os.cpu_count() or 1, 8)
        c = run_oracle(data, val, rounds=rounds, max_iterations=400, threads=cores)
        out["cpu"], out["cpu_threads"] = round(c["dice"], 6), cores
        out["abs_delta_fp32"] = round(abs(h32["dice"] - c["dice"]), 6)
        out["abs_delta_bf16"] = round(abs(h16["dice"] - c["dice"]), 6)
        out["round1_last_loss"] = {"cpu": round(c["losses"][0], 6), "hip_fp32": round(h32["losses"][0], 6),
                                   "hip_bf16": round(h16["losses"][0], 6)}
    fx = os.path.join(ROOT, "tests", "golden", "g19_minifed_dice.npz")
    if os.path.exists(fx):
        g = np.load(fx)
        if int(g["rounds"]) == rounds:
            d = g["dice"]
            out["reference_own_spread"] = {"min": round(float(d.min()), 6), "max": round(float(d.max()), 6),
                                           "mean": round(float(d.mean()), 6), "runs": int(d.size),
                                           "what": "the reference's own classes, 8 dropout seeds x {8, 1} CPU threads "
                                                   "(tests/golden/g19_minifed_dice.npz, oracle/gen_golden.py)"}
    out["seconds"] = round(time.perf_counter() - t0, 1)
    return out


def _conv_symbol(key, dtype_name):
    """The __global__ template csrc/conv_api.hip's measured per-layer rule launches for a profiled conv shape (kind, dtype, N, H, W,
    Cin, Cout, k[, "fused"]) -- a restatement of conv_fwd_impl's dispatch for the shapes of the FedICRA iteration, for the bench
    line only (tools/kbench2.py switches the forms in-process when the rule itself is in question)."""
    kind = key[0]
    try:
        n, h, w, cin, cout, ks = (int(x) for x in key[2:8])
    except (ValueError, TypeError):
        return kind
    fused = len(key) > 8 and key[8] == "fused"
    if kind == "conv_wgrad":
        return "conv_wgrad_rows_kernel" if min(cin, cout) < 32 else ("conv_wgrad_rows64_kernel" if ks == 3 and h >= 64 else "conv_wgrad_quad_kernel")
    if kind not in ("conv_fwd", "conv_dgrad") or dtype_name == "fp32":
        return {"conv_fwd": "conv_fwd_kernel", "conv_dgrad": "conv_fwd_kernel"}.get(kind, kind)
    if ks != 3:
        return "conv_fwd_kernel"
    if cin <= 4 and cout in (8, 16):
        return "conv_narrow_in_kernel"
    if cin <= 32 and cout <= 32 and cin >= 16:
        return "conv_thin_kernel"
    tiles16 = n * -(-h // 16) * -(-w // 16)
    if fused and cout in (32, 64) and 78336 + cin * cout * 18 + 768 <= 160 * 1024 and n * -(-h // 32) * -(-w // 16) >= 1024:
        return "conv_fwd_ws2_kernel (filter resident)"
    if fused and cout % 128 == 0 and cin % 64 == 0 and tiles16 * (cout // 128) >= 1024:
        return "conv_fwd_ws2_kernel"
    if fused and cout == 64 and cin % 32 == 0 and n * -(-h // 32) * -(-w // 16) >= 1024:
        return "conv_fwd_ws2_kernel (32-row tiles)"
    if cin >= 32 and cout >= 32:
        return "conv_fwd_ws_kernel"
    return "conv_fwd_kernel"


def roofline_pass(client, a, dtype_name):
    """Eager, instrumented iterations: HIP events around every C-ABI launch on the launch stream."""
    from fedicra_amd import _lib as L
    client.use_graph = False
    # every launch timed ALONE: the timed rounds run the K-1 LC forwards on a second stream beside the client's own forward
    # (flower_pCE_2D.MyClient.probe_beside), where two kernels share the chip and an event pair measures both
    client.probe_beside = False
    iters = a.round_iters                                # the timed mix: round_iters - 3 head-phase + 3 body-phase iterations
    client.args.iters = iters
    cfg = {"iter_global": 60, "iters": iters, "eval_iters": 10 * iters, "batch_size": a.batch, "stage": "fit"}
    client.args.iters = 4
    L.profile_begin(subtract_overhead=False)             # (counted, not priced: the launches a rocprofv3 run of this process sees too)
    client._train(dict(cfg, iters=4))                    # warm the eager path (both phases: 1 head + 3 body iterations)
    warm = L.profile_end().summary()
    client.args.iters = iters
    L.profile_begin(subtract_overhead=False)
    client._train(cfg)
    kp = L.profile_end()
    prof = kp.summary()
    total_ms = sum(v["ms"] for v in prof.values())
    fam_of = lambda k: "conv_fwd" if k[0] == "conv_dgrad" else k[0]
    # every launch of THIS PROCESS per family (warm-up + instrumented iterations): the population a PMC pass over
    # `bench.py --roofline-only` counts, so that tools/pmc_traffic.py can divide totals by totals (the per-launch
    # means of the 10 instrumented iterations and of the 14 a counter pass sees are different head : body mixes)
    process_totals = {}
    for part in (warm, prof):
        for k, v in part.items():
            t = process_totals.setdefault(fam_of(k), {"launches": 0, "algorithmic_bytes": 0.0, "algorithmic_flops": 0.0})
            t["launches"] += v["calls"]
            t["algorithmic_bytes"] += v["bytes"]
            t["algorithmic_flops"] += v["flops"]
    # dominant kernel = the kernel FAMILY (one __global__ template: conv_fwd also serves dgrad) with the largest
    # share of GPU time; its launches are priced together: achieved = sum(algorithmic work) / sum(duration),
    # i.e. per-launch algorithmic work / average launch duration.
    fam = {}
    for k, v in prof.items():
        name = "conv_fwd" if k[0] == "conv_dgrad" else k[0]
        f = fam.setdefault(name, {"calls": 0, "ms": 0.0, "flops": 0.0, "bytes": 0.0})
        for q in ("calls", "ms", "flops", "bytes"):
            f[q] += v[q]
    fname, dom = max(fam.items(), key=lambda kv: kv[1]["ms"])
    # per-launch min-roofline: each (kind, shape) is priced against whichever of MFMA peak / HBM peak bounds it;
    # frac = sum of those ideal times / sum of measured times (SURVEY section 8d "per-layer min(MFMA, HBM)")
    pk_f, pk_b = MFMA_PEAK[dtype_name] * 1e12, HBM_PEAK_GBS * 1e9

    def ideal_ms(v, executed=True):
        # an algebraic form (the statistics-only head from the input's autocorrelation, csrc/xcorr.hip) is priced with the
        # work it EXECUTES: crediting it with the convolution it replaces made that one launch "0.90 of the MFMA peak"
        # and lifted the all-conv figure by 0.03; the reference-credited variant is printed beside it
        f = v.get("xflops", v["flops"]) if executed else v["flops"]
        return max(f / pk_f, v["bytes"] / pk_b) * 1e3

    fam_keys = [k for k in prof if (("conv_fwd" if k[0] == "conv_dgrad" else k[0]) == fname)]
    conv_keys = [k for k in prof if k[0].startswith("conv")]
    min_roof_fam = sum(ideal_ms(prof[k]) for k in fam_keys) / max(sum(prof[k]["ms"] for k in fam_keys), 1e-9)
    min_roof_conv = sum(ideal_ms(prof[k]) for k in conv_keys) / max(sum(prof[k]["ms"] for k in conv_keys), 1e-9)
    min_roof_conv_ref = sum(ideal_ms(prof[k], False) for k in conv_keys) / max(sum(prof[k]["ms"] for k in conv_keys), 1e-9)
    # the K-1 batched LC forwards (84-image launches): the chain that is the critical path of every iteration
    gi = a.batch * (FEDERATION - 1)
    probe_keys = [k for k in conv_keys if len(k) > 2 and k[2] == gi]
    min_roof_probe = sum(ideal_ms(prof[k]) for k in probe_keys) / max(sum(prof[k]["ms"] for k in probe_keys), 1e-9)
    probe_ms_per_step = sum(prof[k]["ms"] for k in probe_keys) / float(iters)
    ideal_ms_per_step = sum(ideal_ms(v) for v in prof.values()) / float(iters)
    alg_bytes_per_step = sum(v["bytes"] for v in prof.values()) / float(iters)
    calls = dom["calls"]
    avg_ms = dom["ms"] / calls
    flops, nbytes = dom["flops"] / calls, dom["bytes"] / calls
    ai = flops / max(nbytes, 1.0)
    mf_peak = MFMA_PEAK[dtype_name]
    ridge = mf_peak * 1e12 / (HBM_PEAK_GBS * 1e9)
    if flops > 0 and ai >= ridge:
        bound, ach, peak, unit = "mfma", flops / (avg_ms * 1e-3) / 1e12, mf_peak, "TFLOP/s"
    else:
        bound, ach, peak, unit = "hbm", nbytes / (avg_ms * 1e-3) / 1e9, HBM_PEAK_GBS, "GB/s"
    # the single heaviest launch shape of the dominant family and the __global__ symbol csrc/conv_api.hip dispatches it to
    # ("name the single heaviest symbol and its own frac beside the family")
    dk = max(fam_keys, key=lambda k: prof[k]["ms"])
    dv = prof[dk]
    d_us = dv["ms"] / dv["calls"] * 1e3
    d_ideal_us = ideal_ms(dv) / dv["calls"] * 1e3
    dominant_symbol = {"symbol": _conv_symbol(dk, dtype_name), "shape": "/".join(map(str, dk)), "launches_per_step": dv["calls"] / float(iters),
                       "avg_us": round(d_us, 1), "ideal_us": round(d_ideal_us, 1), "frac": round(d_ideal_us / max(d_us, 1e-9), 4),
                       "bound": "mfma" if dv.get("xflops", dv["flops"]) / pk_f >= dv["bytes"] / pk_b else "hbm",
                       "share_of_gpu_time": round(dv["ms"] / total_ms, 4)}
    breakdown = {}
    for k, v in prof.items():
        breakdown[k[0]] = breakdown.get(k[0], 0.0) + v["ms"]
    conv_flops = sum(v["flops"] for k, v in prof.items() if k[0].startswith("conv"))
    # what the launches EXECUTE: differs from the reference-defined count where an algebraic form stands in for a layer (the
    # statistics-only head of the LC forwards from the input's autocorrelation, csrc/xcorr.hip: 13 x 64 x 64 instead of
    # 9 x 64 x 512 multiply-adds per pixel) -- a fraction of peak that rises because work was REMOVED shows here
    conv_xflops = sum(v.get("xflops", v["flops"]) for k, v in prof.items() if k[0].startswith("conv"))
    roof = {"bound": bound, "achieved": round(ach, 2), "peak": peak, "unit": unit, "frac": round(ach / peak, 4),
            "traffic": None, "kernel": f"{fname} (all shapes of the FedICRA iteration)/{dtype_name}",
            "profile_command": "python bench.py --roofline-only  (profiles/*_roofline_kernel_stats.csv, *_pmc_traffic.json)",
            "dominant_symbol": dominant_symbol,
            "launches_per_step": calls / float(iters), "avg_us": round(avg_ms * 1e3, 2),
            "arithmetic_intensity_flop_per_byte": round(ai, 1),
            "frac_of_mfma_peak": round(flops / (avg_ms * 1e-3) / 1e12 / mf_peak, 4),
            "algorithmic_flops_per_launch": flops, "algorithmic_bytes_per_launch": nbytes,
            "share_of_gpu_time": round(dom["ms"] / total_ms, 4),
            "min_roofline_frac": round(min_roof_fam, 4), "min_roofline_frac_all_conv": round(min_roof_conv, 4),
            "min_roofline_frac_all_conv_reference_flops": round(min_roof_conv_ref, 4),
            "min_roofline_frac_probe_chain": round(min_roof_probe, 4), "probe_chain_ms_per_step": round(probe_ms_per_step, 4),
            "ideal_ms_per_step": round(ideal_ms_per_step, 4), "algorithmic_bytes_per_step": alg_bytes_per_step,
            "min_roofline_note": "sum over launches of max(EXECUTED flops / MFMA peak, bytes / HBM peak) / sum of measured durations "
                                 "(an algebraic form is priced with the work it executes; ..._reference_flops credits it with the "
                                 "layer it replaces); probe_chain = the 84-image launches of the K-1 LC forwards; "
                                 "per-shape table: profiles/*_per_layer_roofline.txt",
            "process_totals": process_totals,
            "instrumented_iterations": f"{iters - 3} head-phase + 3 body-phase (the timed mix), LC forwards in line: every launch timed alone",
            "hip_launches_per_step": round(sum(v["calls"] for v in prof.values()) / float(iters), 1),
            "conv_flops_per_step": conv_flops / float(iters),
            "executed_flops_per_step": conv_xflops / float(iters),
            "executed_flops_note": "conv_flops_per_step = the reference-defined algorithmic conv FLOPs of the iteration (what "
                                   "frac_of_mfma_peak divides); executed_flops_per_step = what the launches really multiply",
            "kernel_time_breakdown_ms_per_step": {k: round(v / float(iters), 4) for k, v in sorted(breakdown.items())}}
    # HBM traffic cannot be counted from inside the process: it comes from the committed rocprofv3 PMC run of this same
    # workload (FETCH_SIZE / WRITE_SIZE in separate passes, gfx950 correction applied -- see the json's header)
    # Only a file taken on THIS build counts: the json records the kernel-source hash of the tree it was measured in
    # (fedicra_amd._lib.source_hash -- the GPU box has no .git); any other file is refused and `traffic` stays null.
    import glob
    here = L.source_hash()
    roof["build_source_hash"] = here
    found = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_traffic.json")), key=os.path.getmtime)
    refused = []
    for pmc_path in reversed(found):
        try:
            d = json.load(open(pmc_path))
            if d.get("workload_key") != workload_key(a, dtype_name):
                continue
            if d.get("source_hash") != here:
                refused.append(os.path.basename(pmc_path))
                continue
            fam_pmc = d["families"].get(fname)
            if fam_pmc and "hbm_bytes_per_launch" in fam_pmc:
                roof["traffic"] = fam_pmc["hbm_bytes_per_launch"]
                # totals over the SAME launch population (every launch of the counted process), not a ratio of two means
                roof["traffic_over_algorithmic"] = fam_pmc.get("traffic_over_algorithmic")
                roof["traffic_population"] = fam_pmc.get("population")
                roof["traffic_head"] = d.get("source_hash")
                roof["traffic_source"] = "profiles/%s (rocprofv3 --pmc, bytes per launch)" % os.path.basename(pmc_path)
                m = fam_pmc.get("mfma")
                if m:                                  # SQ counters of the same family (its own PMC pass): matrix-pipe busy share
                    roof["mfma_busy_pct"] = m.get("mfma_busy_pct")
                    roof["mfma_counters_per_launch"] = {k: m.get(k) for k in ("mfma_busy_cycles_per_launch", "gui_active_cycles_per_launch",
                                                                              "sq_busy_cycles_per_launch", "mops_flops_per_launch")}
                break
        except (OSError, ValueError, KeyError):
            continue
    if roof["traffic"] is None and refused:
        roof["traffic_refused"] = f"taken on another build (source hash differs from {here}): " + ", ".join(refused[:3])
    table = os.environ.get("FEDICRA_BENCH_TABLE")
    if table:
        with open(table, "w") as f:
            f.write("# per-launch-shape roofline of one FedICRA round's local training (bench.py eager instrumented pass, HIP events, "
                    f"{iters - 3} head + 3 body iterations, {dtype_name}); ideal = max(flops / {MFMA_PEAK[dtype_name]} TF/s, bytes / 8 TB/s)\n")
            f.write(f"# {'kind/shape (dtype,N,H,W,Cin,Cout,k)':58s} {'calls':>5s} {'avg_us':>9s} {'ideal_us':>9s} {'frac':>6s} {'TF/s':>8s} {'GB/s':>8s} bound\n")
            for k, v in sorted(prof.items(), key=lambda kv: -kv[1]["ms"]):
                us = v["ms"] / v["calls"] * 1e3
                idl = ideal_ms(v) / v["calls"] * 1e3
                xf = v.get("xflops", v["flops"])
                bound = "mfma" if xf / pk_f >= v["bytes"] / pk_b else "hbm"
                f.write(f"{'/'.join(map(str, k)):60s} {v['calls']:5d} {us:9.1f} {idl:9.1f} {idl / max(us, 1e-9):6.3f} "
                        f"{xf / v['calls'] / (us * 1e-6) / 1e12:8.1f} {v['bytes'] / v['calls'] / (us * 1e-6) / 1e9:8.1f} {bound}\n")
            f.write(f"# TF/s and ideal_us from the flops a launch EXECUTES (conv_stats_xcorr: the autocorrelation form, not the convolution it replaces)\n")
            f.write(f"# family {fname}: min-roofline frac {min_roof_fam:.4f}; all conv launches: {min_roof_conv:.4f} "
                    f"(crediting algebraic forms with the replaced layer: {min_roof_conv_ref:.4f}); probe chain (84-image launches): "
                    f"{min_roof_probe:.4f}, {probe_ms_per_step:.3f} ms per step; ideal {ideal_ms_per_step:.3f} ms per step\n")
    if os.environ.get("FEDICRA_BENCH_VERBOSE"):
        top = sorted(prof.items(), key=lambda kv: -kv[1]["ms"])[:60]
        for k, v in top:
            us = v["ms"] / v["calls"] * 1e3
            print(f"# {'/'.join(map(str, k)):60s} calls {v['calls']:3d} avg {us:8.1f} us  "
                  f"{v['flops'] / v['calls'] / (us * 1e-6) / 1e12:7.2f} TF/s  "
                  f"{v['bytes'] / v['calls'] / (us * 1e-6) / 1e9:8.1f} GB/s", file=sys.stderr)
    client.use_graph = not a.no_graph
    return roof


def workload_key(a, dtype_name):
    return f"c3-unet_lc-{a.batch}x{a.in_chns}x{a.size}-{dtype_name}"


def self_spawn(a):
    """`python bench.py --gpus N` without a launcher: start the N ranks the way the driver would (one per GPU, RCCL)."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(a.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--windows", type=int, default=5,
                    help="timed windows of --steps steps each, back to back after ONE warm-up; value = the median window")
    ap.add_argument("--no-dice", action="store_true", help="skip the Dice-vs-CPU-reference leg (config.dice)")
    ap.add_argument("--workload", default="c3", choices=["c3", "c4", "c5"],
                    help="c3 (default): BASELINE configs[2], the configuration the metric is quoted on; c4: configs[3], the 3D path "
                         "(4 clients, unet_3D, 2x1x128^3 bf16) with the same JSON shape in volumes/s")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp16", "fp32"])
    ap.add_argument("--size", type=int, default=None)
    ap.add_argument("--batch", type=int, default=None)
    ap.add_argument("--in-chns", type=int, default=3, choices=[1, 3])
    ap.add_argument("--round-iters", type=int, default=10)
    ap.add_argument("--loader-batches", type=int, default=8,
                    help="training batches resident per client = len(trainloader) = batches of one ALA epoch")
    ap.add_argument("--data", default="host", choices=["host", "resident"],
                    help="where the training batches live: pinned host memory, staged over PCIe beside the compute stream "
                         "inside the timed region (default; the reference's DataLoader), or resident in HBM")
    ap.add_argument("--no-resident", action="store_true", help="skip the resident-data rate reported beside the headline")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-fp32", action="store_true", help="skip the fp32 parity-mode rate reported beside the headline")
    ap.add_argument("--no-3d", action="store_true", help="skip the configs[3] / configs[4] legs (config.c4 / config.c5)")
    ap.add_argument("--no-rccl", action="store_true",
                    help="N = 1 only: do not create the single-rank RCCL process group through which the round's weighted "
                         "all-reduce is issued (config.rccl_single_rank)")
    ap.add_argument("--roofline-only", action="store_true",
                    help="only the instrumented eager iterations of the roofline object (the command rocprofv3 is pointed at "
                         "for profiles/*_roofline_kernel_stats.csv and the PMC traffic passes: same launch mix)")
    a = ap.parse_args()
    a.size = a.size or (512 if a.workload == "c3" else 128)
    a.batch = a.batch or (12 if a.workload == "c3" else 2)
    a.classes = 2 if a.in_chns == 1 else 3

    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_spawn(a))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    # GPU_MAX_HW_QUEUES is left at the runtime's default (4).  A rank drives five HIP streams once a communicator exists, and eight
    # hardware queues measured 0.3-0.6 ms off the ALA epoch -- but WHICH streams then share a queue follows from the order the
    # streams were made in, and one order in three put the batch staging behind the compute queue: 66.3 -> 72 ms of training per round
    # (same box, gpurun_out/v8_matrix.log: 8 queues without the RCCL group 1 304 / 1 318 images/s against 1 378-1 390 for 4 queues with
    # or without it and 8 with it; a form of the step with one compute stream: 1 195).  Four queues measured the same either way.
    from fedicra_amd.comm import init_process_group_from_env
    rank, local, world = init_process_group_from_env()
    assert torch.cuda.is_available(), "bench.py needs an MI355X"
    assert world == a.gpus, f"WORLD_SIZE {world} != --gpus {a.gpus}"
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)
    import torch.distributed as dist

    from fedicra_amd import _lib, streams
    _lib.lib()
    streams.init(dev)              # every HIP stream of this rank, in one fixed order, before the communicator brings its own
    # N = 1: the round's exchange still goes through RCCL -- a process group of ONE rank (the sum over one rank is the identity,
    # the result is bit-identical to the no-group path: tests/test_round5_gpu.py), so that the communicator, the side stream,
    # the event fence and their interplay with the captured steps execute on the one GPU the driver's N = 1 run has
    a.rccl_single_rank, rccl_note = False, None
    if world == 1 and a.workload == "c3" and not a.no_rccl and not a.roofline_only:
        try:
            with socket.socket() as sk:
                sk.bind(("127.0.0.1", 0))
                port = sk.getsockname()[1]
            with _stdout_to_stderr():
                dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1)
                probe = torch.ones(8, device=dev)
                dist.all_reduce(probe)
                torch.cuda.synchronize()
            assert float(probe.sum().item()) == 8.0
            a.rccl_single_rank = True
        except Exception as e:  # noqa: BLE001 -- the headline number must not depend on it; the line says what happened
            rccl_note = repr(e)[:300]
            if dist.is_initialized():
                dist.destroy_process_group()
    if a.workload in ("c4", "c5"):
        main_c4(a, rank, local, world, dev, dist)
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return
    assert world <= FEDERATION, "configs[2] is a federation of 8 clients, one per GPU"

    fed = Federation(a, rank, world, dev, a.dtype)
    if a.roofline_only:
        roof = roofline_pass(fed.client, a, a.dtype)
        if rank == 0:
            print(json.dumps({"roofline": roof}), flush=True)
        return
    # value = the MEDIAN of --windows (5) back-to-back windows of exactly --steps steps each (one 20-step window is 0.2 s: a
    # +-1 % kernel change is inside its noise); every window's rate is printed beside it
    elapsed, agg_ms, win_s = fed.timed(a.warmup, a.steps, dist, a.windows)
    value = a.steps * a.batch * world / elapsed
    win_rates = [round(a.steps * a.batch * world / w, 2) for w in win_s]
    split = fed.round_split()
    h2d_bytes_per_step = None
    if a.data == "host":
        h2d_bytes_per_step = a.batch * (a.in_chns * 4 + 1) * a.size * a.size

    resident = None
    if a.data == "host" and not a.no_resident:
        fedr = Federation(a, rank, world, dev, a.dtype, data="resident")
        kr = min(a.steps, a.round_iters)
        elr, aggr, _ = fedr.timed(max(a.warmup, 2 * a.round_iters), kr, dist)     # (two rounds: the ALA epoch's capture is behind it)
        resident = {"images_per_sec": round(kr * a.batch * world / elr, 2), "steps": kr, "ms_per_aggregation_round": round(aggr, 3),
                    "round_split_ms": fedr.round_split()}
        del fedr
        torch.cuda.empty_cache()

    fp32 = None
    if a.dtype != "fp32" and not a.no_fp32:
        # the reference's own arithmetic (--amp 0): exact-fp32 MFMA parity mode, same rounds, shorter sample
        fed32 = Federation(a, rank, world, dev, "fp32")
        k32 = min(a.steps, a.round_iters)
        el32, agg32, _ = fed32.timed(max(a.warmup, 2 * a.round_iters), k32, dist)
        fp32 = {"images_per_sec": round(k32 * a.batch * world / el32, 2), "steps": k32,
                "ms_per_aggregation_round": round(agg32, 3)}
        del fed32
        torch.cuda.empty_cache()

    if rank == 0:
        line = {
            "metric": "images/sec/client (FedICRA local training, 2D U-Net 512^2) ; ms/aggregation round in config",
            "value": round(value, 2), "unit": "images/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(elapsed / a.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": {"bf16": "bf16", "fp16": "f16", "fp32": "f32"}[a.dtype], "data": "synthetic",
            "config": {"workload": f"BASELINE.json configs[2]: {FEDERATION} clients FedICRA, unet_lc (client_num 8), "
                                   f"{a.batch}x{a.in_chns}x{a.size}x{a.size} per batch, {a.classes} classes, one client per "
                                   f"MI355X ({world} hosted); step = one local iteration (fwd, pCE, 7 no-grad LC forwards, bwd "
                                   f"under the freeze schedule, AdamW); round = {a.round_iters} steps + aggregation round "
                                   f"(weighted all-reduce + set_weights with one ALA epoch over {a.loader_batches} batches), "
                                   f"all timed",
                       "clients_hosted": world, "federation": FEDERATION, "global_batch": a.batch * world,
                       "images_per_sec_per_client": round(value / world, 2),
                       "warmup_steps_run": fed.warmup_run,
                       "warmup_note": "--warmup rounded up to whole rounds, at least two (captures of both freeze phases and of "
                                      "the ALA epoch lie behind them)",
                       "value_is": f"median of {len(win_rates)} back-to-back timed windows of {a.steps} steps each",
                       "value_windows": win_rates, "value_min": min(win_rates), "value_max": max(win_rates),
                       "ms_per_aggregation_round": round(agg_ms, 3),
                       "round_split_ms": split,
                       "ala_batches_per_round": a.loader_batches,
                       "data_location": ("pinned host memory; batch i+1 crosses PCIe on a side stream while iteration i computes "
                                         "(inside the timed region)") if a.data == "host" else "resident in HBM",
                       "h2d_bytes_per_step": h2d_bytes_per_step,
                       # the process's HIP streams by role and position in the creation order (fedicra_amd/streams.py)
                       "streams": _stream_positions(),
                       "resident_images_per_sec": None if resident is None else resident["images_per_sec"],
                       "resident": resident,
                       "fp32_images_per_sec": None if fp32 is None else fp32["images_per_sec"],
                       "fp32": fp32, "hipgraph": not a.no_graph, "parallelism": f"fed-dp{world}"},
        }
        if world > 1:
            line["config"].update({"rccl_ranks": world if fed.backend == "nccl" else 0, "dist_backend": fed.backend,
                                   "allreduce_us_per_round": round(split["collective"] * 1e3, 1),
                                   "overlap_hidden_ms": split["overlap_hidden"]})
        else:
            line["config"]["rccl_single_rank"] = {
                "executed": bool(a.rccl_single_rank), "backend": dist.get_backend() if dist.is_initialized() else None,
                "all_reduce_calls_issued": fed.agg.collectives_issued, "error": rccl_note,
                "allreduce_us_per_round": round(split["collective"] * 1e3, 1),
                "what": "the round's weighted all-reduce (flat fp32 state + int64 counters) issued through a ONE-rank RCCL "
                        "process group on the side stream, event-fenced against the training stream, inside the timed rounds"}
        if not a.no_roofline:
            try:
                roof = roofline_pass(fed.client, a, a.dtype)
                line["roofline"] = roof
                step_s = (elapsed - agg_ms * 1e-3 * len(fed.agg_events)) / a.steps      # training part of a step
                tf = roof["conv_flops_per_step"] / step_s / 1e12          # same head : body mix on both sides of the ratio
                line["config"]["conv_tflops_per_gpu"] = round(tf, 2)
                line["config"]["frac_of_mfma_peak"] = round(tf / MFMA_PEAK[a.dtype], 4)
                line["config"]["frac_of_mfma_peak_is"] = ("reference-defined conv FLOPs of a step / time of the TRAINING steps "
                                                          "(the aggregation rounds' time excluded); whole-round figure beside it")
                tf_round = roof["conv_flops_per_step"] * a.steps / elapsed / 1e12    # the ALA epoch's time in, its conv work not counted
                line["config"]["frac_of_mfma_peak_whole_round"] = round(tf_round / MFMA_PEAK[a.dtype], 4)
                line["config"]["executed_tflops_per_gpu"] = round(roof["executed_flops_per_step"] / step_s / 1e12, 2)
                line["config"]["executed_frac_of_mfma_peak"] = round(roof["executed_flops_per_step"] / step_s / 1e12 / MFMA_PEAK[a.dtype], 4)
            except Exception as e:  # noqa: BLE001  (never lose the headline number to the instrumented pass)
                line["roofline"] = {"error": repr(e)}
        if world == 1 and not a.no_3d:
            # configs[3] / configs[4] in front of the driver: short legs outside the c3 timed region, on the
            # single-GPU line only (a multi-GPU launch measures the c3 federation's scaling and nothing else)
            del fed
            torch.cuda.empty_cache()
            for kind in ("c4", "c5"):
                try:
                    line["config"][kind] = volumes_leg(a, rank, world, dev, dist, kind)
                except Exception as e:  # noqa: BLE001
                    line["config"][kind] = {"error": repr(e)[:300]}
        if world == 1 and not a.no_dice:
            try:
                line["config"]["dice"] = dice_leg(with_cpu=not a.no_cpu_baseline)
            except Exception as e:  # noqa: BLE001
                line["config"]["dice"] = {"error": repr(e)[:300]}
        if world == 1 and not a.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(a)
    if world > 1:
        dist.barrier()
    if dist.is_initialized():
        with _stdout_to_stderr():                        # (whatever the communicator says on its way out is not the result)
            dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(line), flush=True)              # the ONE line of stdout, and the last thing written to it


if __name__ == "__main__":
    main()
