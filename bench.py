#!/usr/bin/env python
"""bench.py -- images/s of the FedICRA local-training hot path on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Workload = BASELINE.json configs[1]: FedAvg clients, 2D U-Net(in=1, classes=2), 12x1x256x256 synthetic
slices, one client per GPU.  A *step* is one local training iteration (zero-grad, forward, pCE loss,
backward, AdamW, poly-LR) on one batch of 12 images whose data is already resident in HBM; every
``--round-iters`` (10, the reference default, flower_runner.py:38-39) steps form a federated round that ends
with the weighted parameter aggregation (RCCL all-reduce over xGMI when N > 1) and the load of the global
weights -- all inside the timed region.  value = images/s summed over all clients (weak scaling: per-GPU
work is fixed).  Rank 0 prints ONE JSON line.

Extra objects on the line (see DESIGN.md "measurement"):
  roofline     -- the dominant kernel (largest share of GPU time) priced against its roofline; durations
                  measured live with HIP events around each launch in a separate, eager, instrumented pass.
  cpu_baseline -- the CPU oracle (oracle/, kind "port") timed on this box's host cores on a bounded sample
                  of the same workload (rank 0, N == 1 only).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: HBM3E 8 TB/s
MFMA_PEAK = {"bf16": 2500.0, "fp16": 2500.0, "f32": 157.3}   # TFLOP/s dense (guide: 2.5 PF bf16, 157.3 TF fp32-input MFMA)

# conv-only algorithmic FLOPs per image, UNet(1,2): F_train = 3*F_fwd - first-layer dgrad
F_FWD_256 = 5.8615e9


def f_train(size, in_chns=1):
    f_fwd = F_FWD_256 * (size / 256.0) ** 2
    return 3.0 * f_fwd - 2.0 * in_chns * 16 * 9 * size * size


def make_args(a, cid, nclients):
    return argparse.Namespace(strategy="FedAvg", amp=0, model="unet", cid=cid, min_num_clients=nclients,
                              num_classes=2, img_class="faz", base_lr=0.01, max_iterations=30000,
                              iters=a.round_iters, rep_iters=3, alpha=0.5, snapshot_path=None,
                              use_graph=not a.no_graph)


def device_loader(n_batches, batch, size, cid, device):
    from fedicra_amd.synth import phantom_batch
    out = []
    for i in range(n_batches):
        img, weak, _ = phantom_batch(batch, size, 1, 2, cid=cid, index=i)
        out.append({"image": torch.from_numpy(img).to(device), "label": torch.from_numpy(weak).to(device)})
    return out


def cpu_baseline(a):
    """The oracle's local_train on the host cores: bounded sample of the same workload."""
    from oracle import fed_ref
    from oracle.unet_ref import RefUNet
    from fedicra_amd.synth import phantom_batch
    # torch's CPU conv path stops scaling (and collapses from oversubscription) far below the 256 hardware
    # threads of the GPU box's host: use at most 32 threads and report that number as `cores`.
    cores = min(os.cpu_count() or 1, int(os.environ.get("FEDICRA_CPU_THREADS", "32")))
    torch.set_num_threads(cores)
    torch.manual_seed(2022)
    m = RefUNet(1, 2)
    batches = []
    for i in range(2):
        img, weak, _ = phantom_batch(a.batch, a.size, 1, 2, cid=0, index=i)
        batches.append({"image": torch.from_numpy(img), "label": torch.from_numpy(weak)})
    st = fed_ref.TrainState(0.01)
    fed_ref.local_train(m, st, batches, iters=1, num_classes=2, base_lr=0.01, max_iterations=30000)   # warm-up
    n, t0 = 0, time.perf_counter()
    while n < 2 or (time.perf_counter() - t0 < 12.0 and n < 40):
        fed_ref.local_train(m, st, batches, iters=1, num_classes=2, base_lr=0.01, max_iterations=30000)
        n += 1
    dt = time.perf_counter() - t0
    return {"value": round(n * a.batch / dt, 3), "unit": "images/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{n} training iterations of {a.batch}x1x{a.size}x{a.size} (oracle.fed_ref.local_train, "
                      f"torch {torch.__version__} CPU fp32) after 1 warm-up"}


def roofline_pass(client, a, dtype_name):
    """Eager, instrumented iterations: HIP events around every C-ABI launch on the launch stream."""
    from fedicra_amd import _lib as L
    client.use_graph = False
    cfg = {"iter_global": 0, "iters": 3, "eval_iters": 10, "batch_size": a.batch, "stage": "fit"}
    client._train(cfg)                                   # warm the eager path
    L.profile_begin()
    client._train(cfg)
    kp = L.profile_end()
    prof = kp.summary()
    L._prof_last_overhead = kp.overhead_ms
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
    key = (fname + " (all shapes of the U-Net step)", dtype_name)
    calls = dom["calls"]
    avg_ms = dom["ms"] / calls
    flops, nbytes = dom["flops"] / calls, dom["bytes"] / calls
    ai = flops / max(nbytes, 1.0)
    mf_peak = MFMA_PEAK[{"bf16": "bf16", "fp16": "fp16"}.get(dtype_name, "f32")]
    ridge = mf_peak * 1e12 / (HBM_PEAK_GBS * 1e9)
    if flops > 0 and ai >= ridge:
        bound, ach, peak, unit = "mfma", flops / (avg_ms * 1e-3) / 1e12, mf_peak, "TFLOP/s"
    else:
        bound, ach, peak, unit = "hbm", nbytes / (avg_ms * 1e-3) / 1e9, HBM_PEAK_GBS, "GB/s"
    breakdown = {}
    for k, v in prof.items():
        b = breakdown.setdefault(k[0], 0.0)
        breakdown[k[0]] = b + v["ms"]
    roof = {"bound": bound, "achieved": round(ach, 2), "peak": peak, "unit": unit, "frac": round(ach / peak, 4),
            "traffic": None, "kernel": "/".join(map(str, key)), "launches_per_step": calls / 3.0,
            "event_bracket_overhead_us_subtracted": round(L._prof_last_overhead * 1e3, 2),
            "avg_us": round(avg_ms * 1e3, 2), "arithmetic_intensity_flop_per_byte": round(ai, 1),
            "frac_of_mfma_peak": round(flops / (avg_ms * 1e-3) / 1e12 / mf_peak, 4),
            "algorithmic_flops_per_launch": flops, "algorithmic_bytes_per_launch": nbytes,
            "share_of_gpu_time": round(dom["ms"] / total_ms, 4),
            "kernel_time_breakdown_ms_per_iter": {k: round(v / 3.0, 4) for k, v in sorted(breakdown.items())}}
    # HBM traffic cannot be counted from inside the process: it comes from the last committed rocprofv3 PMC run of this
    # same workload (FETCH_SIZE / WRITE_SIZE in separate passes, gfx950 correction applied -- see the json's header)
    import glob
    found = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "*_pmc_traffic.json")))
    pmc_path = found[-1] if found else ""                    # the newest committed PMC measurement (tools/pmc_round.sh)
    if dtype_name == "bf16" and a.size == 256 and a.batch == 12 and os.path.exists(pmc_path):
        try:
            fam_pmc = json.load(open(pmc_path))["families"].get(fname)
            if fam_pmc:
                roof["traffic"] = fam_pmc["hbm_bytes_per_launch"]
                roof["traffic_source"] = "profiles/%s (rocprofv3 --pmc, bytes per launch)" % os.path.basename(pmc_path)
        except (OSError, ValueError, KeyError):
            pass
    if os.environ.get("FEDICRA_BENCH_VERBOSE"):
        top = sorted(prof.items(), key=lambda kv: -kv[1]["ms"])[:40]
        for k, v in top:
            us = v["ms"] / v["calls"] * 1e3
            print(f"# {'/'.join(map(str, k)):60s} calls {v['calls']:3d} avg {us:8.1f} us  "
                  f"{v['flops'] / v['calls'] / (us * 1e-6) / 1e12:7.2f} TF/s  "
                  f"{v['bytes'] / v['calls'] / (us * 1e-6) / 1e9:8.1f} GB/s", file=sys.stderr)
    client.use_graph = not a.no_graph
    return roof


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp16", "fp32"])
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--batch", type=int, default=12)
    ap.add_argument("--round-iters", type=int, default=10)
    ap.add_argument("--clients-per-gpu", type=int, default=2,
                    help="clients hosted by each GPU (BASELINE configs[1] = 2 clients); total clients = this x --gpus")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    a = ap.parse_args()

    if a.clients_per_gpu > 2:
        # HIP maps streams onto GPU_MAX_HW_QUEUES hardware queues (default 4); with more than two client streams some
        # would share a queue and serialise (3 clients: 8.1k images/s on 4 queues, 10.2k on 8).  Must be set before the
        # runtime initialises.
        os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
    from fedicra_amd.comm import WeightedAllReduce, init_process_group_from_env
    rank, local, world = init_process_group_from_env()
    assert torch.cuda.is_available(), "bench.py needs an MI355X"
    assert world == a.gpus or world == 1, f"WORLD_SIZE {world} != --gpus {a.gpus}"
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)
    import torch.distributed as dist

    from fedicra_amd import _lib, fl
    from fedicra_amd.flower_common import MyModel
    from fedicra_amd.flower_pCE_2D import MyClient, train_colocated
    from fedicra_amd.networks import net_factory
    from fedicra_amd.networks.unet import set_compute_dtype
    from fedicra_amd.synth import client_num_batches
    _lib.lib()

    # C clients hosted by this rank's GPU (configs[1] has 2 clients: one MI355X holds both), client id = rank*C + j
    Cg = a.clients_per_gpu
    nclients = max(world, 1) * Cg
    all_n = client_num_batches(nclients, a.batch)
    clients, n_local = [], []
    for j in range(Cg):
        cid = rank * Cg + j
        args = make_args(a, cid, nclients)
        torch.manual_seed(2022)                          # the reference seeds every client process with 2022
        net = net_factory(args, net_type="unet", in_chns=1, class_num=2)
        set_compute_dtype(net, a.dtype)
        loader = device_loader(min(all_n[cid], 8), a.batch, a.size, cid, dev)    # resident in HBM before timing starts
        model = MyModel(args, net, loader, loader)
        clients.append(MyClient(args, model, loader, loader))
        n_local.append(all_n[cid])
    client = clients[0]
    agg = WeightedAllReduce(n_local, device=dev)

    def run_steps(nsteps):
        """nsteps local iterations of every hosted client in rounds of round_iters, each round closed by the aggregation."""
        done, agg_ms = 0, []
        while done < nsteps:
            it = min(a.round_iters, nsteps - done)
            cfgs = []
            for c in clients:
                c.args.iters = it
                cfgs.append({"iter_global": done, "iters": it, "eval_iters": 10 * it, "batch_size": a.batch, "stage": "fit"})
            if Cg == 1:
                client._train(cfgs[0])
            else:
                train_colocated(clients, cfgs)           # one HIP stream + captured step per client, interleaved
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            agg.start([c.model.get_device_weights() for c in clients])   # side stream: weighted sum, all-reduce, divide
            for c in clients:
                c.sampled_batches = list(c.trainloader)  # overlapped: next round's batch staging (epoch list)
            glob = agg.finish()
            for c in clients:
                c.model.set_weights(glob, {"iter_global": done})     # FedAvg: plain load of the global state
            e1.record()
            agg_ms.append((e0, e1))
            done += it
        return agg_ms

    run_steps(a.warmup)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    agg_events = run_steps(a.steps)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    agg_ms = sum(e0.elapsed_time(e1) for e0, e1 in agg_events) / max(len(agg_events), 1)

    if rank == 0:
        total_images = a.steps * a.batch * world * Cg
        value = total_images / elapsed
        ft = f_train(a.size)
        conv_tflops_per_gpu = value / world * ft / 1e12
        line = {
            "metric": "images/sec (2D U-Net local training, all clients) ; ms/aggregation round in config",
            "value": round(value, 2), "unit": "images/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(elapsed / a.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": {"bf16": "bf16", "fp16": "f16", "fp32": "f32"}[a.dtype], "data": "synthetic",
            "config": {"workload": f"BASELINE.json configs[1]: {nclients} clients FedAvg, 2D U-Net(1,2) "
                                   f"{a.batch}x1x{a.size}x{a.size} per client, {Cg} client(s) per MI355X (one HIP stream + "
                                   f"captured step each), step = one local iteration of every client, round = "
                                   f"{a.round_iters} steps + weighted aggregation (on-device sum, all-reduce across GPUs)",
                       "clients": nclients, "clients_per_gpu": Cg, "global_batch": a.batch * nclients,
                       "images_per_sec_per_client": round(value / nclients, 2),
                       "ms_per_aggregation_round": round(agg_ms, 4),
                       "conv_tflops_per_gpu": round(conv_tflops_per_gpu, 2),
                       "frac_of_bf16_mfma_peak": round(conv_tflops_per_gpu / MFMA_PEAK["bf16"], 4),
                       "hipgraph": not a.no_graph, "parallelism": f"fed-dp{world}x{Cg}"},
        }
        if not a.no_roofline:
            try:
                line["roofline"] = roofline_pass(client, a, a.dtype)
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
