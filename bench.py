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


def make_args(a, cid):
    return argparse.Namespace(strategy="FedICRA", amp=0, model="unet_lc", cid=cid, min_num_clients=FEDERATION,
                              num_classes=a.classes, img_class="faz" if a.in_chns == 1 else "odoc", base_lr=0.01,
                              max_iterations=30000, iters=a.round_iters, rep_iters=3, alpha=1.0, snapshot_path=None,
                              use_graph=not a.no_graph)


def device_loader(n_batches, a, cid, device):
    from fedicra_amd.synth import phantom_batch
    out = []
    for i in range(n_batches):
        img, weak, _ = phantom_batch(a.batch, a.size, a.in_chns, a.classes, cid=cid, index=i)
        out.append({"image": torch.from_numpy(img).to(device), "label": torch.from_numpy(weak).to(device)})
    return out


class Federation:
    """The hosted client of this rank plus the round loop (fit -> aggregate -> set_weights) around it."""

    def __init__(self, a, rank, world, dev, dtype):
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
        self.loader = device_loader(a.loader_batches, a, cid, dev)      # resident in HBM before timing starts
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
        self.agg = WeightedAllReduce(all_n[cid], device=dev, constant_term=absent)
        if world > 1:
            # communicator creation (seconds, once) must not land inside a timed region, whatever --warmup is
            import torch.distributed as dist
            dist.all_reduce(torch.zeros(1, device=dev if dist.get_backend() == "nccl" else "cpu"))
            torch.cuda.synchronize()
        self.iter_global = 60                            # > 50: the ALA branch runs (flower_common.py:524-526)
        self.agg_events = []

    def run_steps(self, nsteps):
        a, c = self.a, self.client
        done = 0
        while done < nsteps:
            it = min(a.round_iters, nsteps - done)
            c.args.iters = it
            c._train({"iter_global": self.iter_global, "iters": it, "eval_iters": 10 * it, "batch_size": a.batch,
                      "stage": "fit"})
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            self.agg.start(self.model.get_device_weights())      # side stream: pre-scale, all-reduce, divide
            c.sampled_batches = list(c.trainloader)              # overlapped: next round's batch staging (epoch list)
            glob = self.agg.finish()
            self.model.set_weights(glob, {"iter_global": self.iter_global})    # global load + ALA epoch
            e1.record()
            self.agg_events.append((e0, e1))
            self.iter_global += 1
            done += it

    def timed(self, warmup, steps, dist):
        self.run_steps(warmup)
        torch.cuda.synchronize()
        if self.world > 1:
            dist.barrier()
        self.agg_events = []
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
        agg_ms = [e0.elapsed_time(e1) for e0, e1 in self.agg_events]
        return elapsed, sum(agg_ms) / max(len(agg_ms), 1)


def cpu_baseline(a):
    """The oracle's FedICRA local_train (LC forwards included) on the host cores: bounded sample of the same workload."""
    from oracle import fed_ref
    from oracle.unet_ref import RefUNetLC
    from fedicra_amd.synth import phantom_batch
    # torch's CPU conv path stops scaling (and collapses from oversubscription) far below the 256 hardware
    # threads of the GPU box's host: use at most 32 threads and report that number as `cores`.
    cores = min(os.cpu_count() or 1, int(os.environ.get("FEDICRA_CPU_THREADS", "32")))
    torch.set_num_threads(cores)
    torch.manual_seed(2022)
    m = RefUNetLC(a.in_chns, a.classes, 1, FEDERATION, FEDERATION, 0, heads=1)
    img, weak, _ = phantom_batch(a.batch, a.size, a.in_chns, a.classes, cid=0, index=0)
    batches = [{"image": torch.from_numpy(img), "label": torch.from_numpy(weak)}]
    st = fed_ref.TrainState(0.01)
    kw = dict(num_classes=a.classes, base_lr=0.01, max_iterations=30000, img_class="faz" if a.in_chns == 1 else "odoc",
              strategy="FedICRA", rep_iters=0, alpha=1.0, cid=0, num_clients=FEDERATION)
    n, t0 = 0, time.perf_counter()
    while n < 1 or (time.perf_counter() - t0 < 15.0 and n < 20):
        fed_ref.local_train(m, st, batches, iters=1, **kw)       # head-phase iteration: forward, 7 LC forwards, out_conv step
        n += 1
    dt = time.perf_counter() - t0
    return {"value": round(n * a.batch / dt, 3), "unit": "images/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{n} head-phase FedICRA training iteration(s) of {a.batch}x{a.in_chns}x{a.size}x{a.size} incl. the 7 "
                      f"LC forwards (oracle.fed_ref.local_train on RefUNetLC, torch {torch.__version__} CPU fp32), no warm-up"}


def roofline_pass(client, a, dtype_name):
    """Eager, instrumented iterations: HIP events around every C-ABI launch on the launch stream."""
    from fedicra_amd import _lib as L
    client.use_graph = False
    iters = 4
    client.args.iters = iters
    cfg = {"iter_global": 60, "iters": iters, "eval_iters": 10, "batch_size": a.batch, "stage": "fit"}
    client._train(cfg)                                   # warm the eager path
    L.profile_begin(subtract_overhead=False)
    client._train(cfg)                                   # 1 head-phase + 3 body-phase iterations
    kp = L.profile_end()
    prof = kp.summary()
    total_ms = sum(v["ms"] for v in prof.values())
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
    breakdown = {}
    for k, v in prof.items():
        breakdown[k[0]] = breakdown.get(k[0], 0.0) + v["ms"]
    conv_flops = sum(v["flops"] for k, v in prof.items() if k[0].startswith("conv"))
    roof = {"bound": bound, "achieved": round(ach, 2), "peak": peak, "unit": unit, "frac": round(ach / peak, 4),
            "traffic": None, "kernel": f"{fname} (all shapes of the FedICRA iteration)/{dtype_name}",
            "profile_command": "python bench.py --roofline-only  (profiles/*_roofline_kernel_stats.csv, *_pmc_traffic.json)",
            "launches_per_step": calls / float(iters), "avg_us": round(avg_ms * 1e3, 2),
            "arithmetic_intensity_flop_per_byte": round(ai, 1),
            "frac_of_mfma_peak": round(flops / (avg_ms * 1e-3) / 1e12 / mf_peak, 4),
            "algorithmic_flops_per_launch": flops, "algorithmic_bytes_per_launch": nbytes,
            "share_of_gpu_time": round(dom["ms"] / total_ms, 4),
            "conv_flops_per_step": conv_flops / float(iters),
            "kernel_time_breakdown_ms_per_step": {k: round(v / float(iters), 4) for k, v in sorted(breakdown.items())}}
    # HBM traffic cannot be counted from inside the process: it comes from the committed rocprofv3 PMC run of this same
    # workload (FETCH_SIZE / WRITE_SIZE in separate passes, gfx950 correction applied -- see the json's header)
    import glob
    found = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_traffic.json")))
    for pmc_path in reversed(found):
        try:
            d = json.load(open(pmc_path))
            if d.get("workload_key") != workload_key(a, dtype_name):
                continue
            fam_pmc = d["families"].get(fname)
            if fam_pmc:
                roof["traffic"] = fam_pmc["hbm_bytes_per_launch"]
                roof["traffic_source"] = "profiles/%s (rocprofv3 --pmc, bytes per launch)" % os.path.basename(pmc_path)
                break
        except (OSError, ValueError, KeyError):
            continue
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
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp16", "fp32"])
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--batch", type=int, default=12)
    ap.add_argument("--in-chns", type=int, default=3, choices=[1, 3])
    ap.add_argument("--round-iters", type=int, default=10)
    ap.add_argument("--loader-batches", type=int, default=8,
                    help="training batches resident per client = len(trainloader) = batches of one ALA epoch")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-fp32", action="store_true", help="skip the fp32 parity-mode rate reported beside the headline")
    ap.add_argument("--roofline-only", action="store_true",
                    help="only the instrumented eager iterations of the roofline object (the command rocprofv3 is pointed at "
                         "for profiles/*_roofline_kernel_stats.csv and the PMC traffic passes: same launch mix)")
    a = ap.parse_args()
    a.classes = 2 if a.in_chns == 1 else 3

    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_spawn(a))
    from fedicra_amd.comm import init_process_group_from_env
    rank, local, world = init_process_group_from_env()
    assert torch.cuda.is_available(), "bench.py needs an MI355X"
    assert world == a.gpus, f"WORLD_SIZE {world} != --gpus {a.gpus}"
    assert world <= FEDERATION, "configs[2] is a federation of 8 clients, one per GPU"
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)
    import torch.distributed as dist

    from fedicra_amd import _lib
    _lib.lib()

    fed = Federation(a, rank, world, dev, a.dtype)
    if a.roofline_only:
        roof = roofline_pass(fed.client, a, a.dtype)
        if rank == 0:
            print(json.dumps({"roofline": roof}), flush=True)
        return
    elapsed, agg_ms = fed.timed(a.warmup, a.steps, dist)
    value = a.steps * a.batch * world / elapsed

    fp32 = None
    if a.dtype != "fp32" and not a.no_fp32:
        # the reference's own arithmetic (--amp 0): exact-fp32 MFMA parity mode, same rounds, shorter sample
        fed32 = Federation(a, rank, world, dev, "fp32")
        k32 = min(a.steps, a.round_iters)
        el32, agg32 = fed32.timed(a.warmup, k32, dist)
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
                       "ms_per_aggregation_round": round(agg_ms, 3),
                       "ala_batches_per_round": a.loader_batches,
                       "fp32_images_per_sec": None if fp32 is None else fp32["images_per_sec"],
                       "fp32": fp32, "hipgraph": not a.no_graph, "parallelism": f"fed-dp{world}"},
        }
        if not a.no_roofline:
            try:
                roof = roofline_pass(fed.client, a, a.dtype)
                line["roofline"] = roof
                step_s = (elapsed - agg_ms * 1e-3 * len(fed.agg_events)) / a.steps      # training part of a step
                tf = roof["conv_flops_per_step"] / step_s / 1e12
                line["config"]["conv_tflops_per_gpu"] = round(tf, 2)
                line["config"]["frac_of_mfma_peak"] = round(tf / MFMA_PEAK[a.dtype], 4)
            except Exception as e:  # noqa: BLE001  (never lose the headline number to the instrumented pass)
                line["roofline"] = {"error": repr(e)}
        if world == 1 and not a.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(a)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
