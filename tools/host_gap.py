#!/usr/bin/env python
"""What does the HOST do between two graph replays of the c3 training step, and does the GPU wait for it?  (VERDICT r4 item 5:
~300 us of idle GPU at the start of every iteration in the rocprofv3 timeline.)  GPU box:  python tools/host_gap.py
  1. the bench's own rounds (bench.Federation.run_steps), wall time per round and per iteration;
  2. the same rounds with every host-side piece of MyClient.train_steps timed by perf_counter (stage / freeze / replay / history /
     prefetch) -- host milliseconds per iteration, i.e. how far the host runs ahead of a ~6.6 ms GPU iteration;
  3. the captured head-phase step replayed back to back with NOTHING else on the host: the GPU-bound floor of an iteration."""
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
    fed.run_steps(30)
    torch.cuda.synchronize()
    # 1. plain rounds
    t0 = time.perf_counter()
    fed.run_steps(30)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    fed.agg_events, fed.train_events = fed.agg_events[-3:], fed.train_events[-3:]
    sp = fed.round_split()
    print(f"[1] 3 rounds: {el / 3 * 1e3:.2f} ms per round; train {sp['train']:.2f} ms / 10 iterations, ALA {sp['ala']:.2f} ms")
    # 2. host pieces
    c = fed.client
    acc = {}

    def wrap(obj, name, key):
        f = getattr(obj, name)

        def g(*x, **k):
            t = time.perf_counter()
            try:
                return f(*x, **k)
            finally:
                acc[key] = acc.get(key, 0.0) + time.perf_counter() - t
        setattr(obj, name, g)
        return f

    saved = [(c, "_stage", wrap(c, "_stage", "stage")), (c, "_set_freeze", wrap(c, "_set_freeze", "set_freeze")),
             (c, "_prefetch_next", wrap(c, "_prefetch_next", "prefetch"))]
    for rec in c._steps.values():
        if rec.graph is not None:
            saved.append((rec.graph, "replay", wrap(rec.graph, "replay", "graph.replay")))
    t_gen = 0.0
    iters = 0
    for _ in range(3):
        cfg = {"iter_global": fed.iter_global, "iters": 10, "eval_iters": 100, "batch_size": a.batch, "stage": "fit"}
        c.args.iters = 10
        gen = c.train_steps(cfg)
        while True:
            t = time.perf_counter()
            try:
                next(gen)
            except StopIteration:
                break
            finally:
                t_gen += time.perf_counter() - t
            iters += 1
        torch.cuda.synchronize()
    for obj, name, f in saved:
        setattr(obj, name, f)
    print(f"[2] host time per iteration (enqueue only, {iters} iterations): total {t_gen / iters * 1e3:.3f} ms; " +
          ", ".join(f"{k} {v / iters * 1e3:.3f}" for k, v in sorted(acc.items())) +
          f"; rest (history writes, bookkeeping) {(t_gen - sum(acc.values())) / iters * 1e3:.3f}")
    # 3. replay-only floor
    for name, rec in c._steps.items():
        if rec.graph is None:
            continue
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        e0.record()
        for _ in range(20):
            rec.graph.replay()
        e1.record()
        t_host = time.perf_counter() - t0
        torch.cuda.synchronize()
        print(f"[3] {name}: 20 back-to-back replays: {e0.elapsed_time(e1) / 20:.3f} ms per replay on the GPU, host enqueue {t_host / 20 * 1e3:.3f} ms per replay")


if __name__ == "__main__":
    main()
