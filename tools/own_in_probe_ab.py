#!/usr/bin/env python
"""Same process, same weights: rounds of the bench's federation with the head-phase own forward as group 0 of the batched LC
forwards (MyClient.own_in_probe) and as a pass of its own, alternately -- training milliseconds per round from HIP events, and the
captured head-phase step replayed alone.  GPU box:  python tools/own_in_probe_ab.py"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch


def main():
    import bench
    a = argparse.Namespace(batch=12, size=512, in_chns=3, classes=3, round_iters=10, loader_batches=8, data="host", no_graph=False,
                           rccl_single_rank=False)
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    fed = bench.Federation(a, 0, 1, dev, "bf16")
    c = fed.client
    for rep in range(2):
        for merged in ((True, False) if os.environ.get("AB_MERGED_FIRST") else (False, True)):
            c.own_in_probe = merged
            c._steps.clear()                       # captured steps belong to the form they were captured in
            fed.run_steps(30)                      # eager, capture, replay
            torch.cuda.synchronize()
            fed.agg_events, fed.train_events = [], []
            t0 = time.perf_counter()
            fed.run_steps(30)
            torch.cuda.synchronize()
            el = (time.perf_counter() - t0) / 3 * 1e3
            sp = fed.round_split()
            # head-phase step alone
            rec = c._steps.get("head")
            alone = None
            if rec is not None and rec.graph is not None:
                for _ in range(3):
                    rec.graph.replay()
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(20):
                    rec.graph.replay()
                e1.record()
                torch.cuda.synchronize()
                alone = e0.elapsed_time(e1) / 20
            print(f"own_in_probe={int(merged)} rep {rep}: {el:.2f} ms per round, train {sp['train']:.2f} ms / 10 iterations, ALA {sp['ala']:.2f}; "
                  f"head-phase step replayed alone {alone if alone is None else round(alone, 3)} ms; merged passes so far "
                  f"{c.__dict__.get('merged_iterations', 0)}", flush=True)


main()
