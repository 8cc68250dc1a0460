#!/usr/bin/env python
"""Probe: aggregate throughput of K clients co-located on ONE MI355X, each replaying its own captured training step on
its own HIP stream (timing only -- the clients still share the process-wide scratch arena here, so numerics are not
meaningful).  python tools/colocate_probe.py [K ...]"""
import argparse
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    ks = [int(v) for v in sys.argv[1:]] or [1, 2, 3, 4]
    a = argparse.Namespace(dtype="bf16", size=256, batch=12, round_iters=10, no_graph=False)
    dev = torch.device("cuda", 0)
    from fedicra_amd.flower_common import MyModel
    from fedicra_amd.flower_pCE_2D import MyClient
    from fedicra_amd.networks import net_factory
    from fedicra_amd.networks.unet import set_compute_dtype
    clients = []
    for k in range(max(ks)):
        args = bench.make_args(a, k, max(ks))
        torch.manual_seed(2022)
        net = net_factory(args, net_type="unet", in_chns=1, class_num=2)
        set_compute_dtype(net, a.dtype)
        loader = bench.device_loader(4, a.batch, a.size, k, dev)
        model = MyModel(args, net, loader, loader)
        c = MyClient(args, model, loader, loader)
        args.iters = 4
        c._train({"iter_global": 0, "iters": 4, "eval_iters": 40, "batch_size": a.batch, "stage": "fit"})
        torch.cuda.synchronize()
        clients.append(c)
    graphs = [c._steps["all"].graph for c in clients]
    streams = [torch.cuda.Stream() for _ in clients]
    reps = 200
    for K in ks:
        for _ in range(2):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for r in range(reps):
                for g, s in zip(graphs[:K], streams[:K]):
                    with torch.cuda.stream(s):
                        g.replay()
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
        print(f"K={K}: {dt / reps * 1e3:.3f} ms per round of K steps, {K * a.batch * reps / dt:.0f} images/s aggregate")
        # host cost of one graph launch (enqueue only), and the same K chains driven by K host threads
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        with torch.cuda.stream(streams[0]):
            for r in range(20):
                graphs[0].replay()
        th = (time.perf_counter() - t0) / 20
        torch.cuda.synchronize()
        import threading

        def work(g, s):
            with torch.cuda.stream(s):
                for r in range(reps):
                    g.replay()
            s.synchronize()

        for _ in range(2):
            torch.cuda.synchronize()
            ts = [threading.Thread(target=work, args=(g, s)) for g, s in zip(graphs[:K], streams[:K])]
            t0 = time.perf_counter()
            for t in ts:
                t.start()
            for t in ts:
                t.join()
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
        print(f"      host enqueue of one graph launch {th * 1e3:.3f} ms; {K} host threads: {dt / reps * 1e3:.3f} ms per round, "
              f"{K * a.batch * reps / dt:.0f} images/s aggregate")


main()
