#!/usr/bin/env python
"""ms per FedICRA aggregation round on one client: load of the global state + one ALA epoch over the client's batches
(flower_common.MyModel.set_weights), eager launches vs the captured iteration.
    python tools/ala_bench.py [--size 256] [--batches 8] [--model unet_lc]"""
import argparse
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--batches", type=int, default=8)
    ap.add_argument("--batch", type=int, default=12)
    ap.add_argument("--in-chns", type=int, default=1)
    ap.add_argument("--graph-only", action="store_true", help="skip the eager leg (profiling runs)")
    ap.add_argument("--rounds", type=int, default=5)
    a = ap.parse_args()
    from fedicra_amd.flower_common import DeviceWeights, MyModel
    from fedicra_amd.networks import net_factory
    from fedicra_amd.networks.unet import set_compute_dtype
    from fedicra_amd.synth import phantom_batch
    dev = torch.device("cuda", 0)
    ncls = 2 if a.in_chns == 1 else 3
    batches = []
    for i in range(a.batches):
        img, weak, _ = phantom_batch(a.batch, a.size, a.in_chns, ncls, cid=2, index=i)
        batches.append({"image": torch.from_numpy(img).to(dev), "label": torch.from_numpy(weak).to(dev)})
    for use_graph in ((True,) if a.graph_only else (False, True)):
        args = argparse.Namespace(strategy="FedICRA", amp=0, model="unet_lc", cid=2, min_num_clients=8, num_classes=ncls,
                                  img_class="faz" if a.in_chns == 1 else "odoc", base_lr=0.01, max_iterations=30000, iters=6,
                                  rep_iters=3, alpha=1.0, snapshot_path=None, use_graph=use_graph)
        torch.manual_seed(2022)
        net = net_factory(args, net_type="unet_lc", in_chns=a.in_chns, class_num=ncls)
        set_compute_dtype(net, "bf16")
        model = MyModel(args, net, batches, batches)
        model.train()
        model.start_phase = False
        sys.stdout = open(os.devnull, "w")
        ts = []
        for r in range(a.rounds):
            glob = DeviceWeights(net.flat_state + 0.01 * torch.randn_like(net.flat_state), net.flat_counters.clone())
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            model.set_weights(glob, {"iter_global": 60 + r})
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) * 1e3)
        sys.stdout = sys.__stdout__
        print(f"{'hipGraph' if use_graph else 'eager   '}: set_weights with one ALA epoch over {a.batches} batches of "
              f"{a.batch}x{a.in_chns}x{a.size}^2: {min(ts[2:]):.2f} ms  ({min(ts[2:]) / a.batches:.2f} ms per batch; first calls "
              f"{ts[0]:.0f}, {ts[1]:.0f} ms)")


main()
