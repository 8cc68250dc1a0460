#!/usr/bin/env python
"""Which host-side statement of MyClient.train_steps costs the GPU ~0.35 ms per iteration?  The captured head-phase step replayed
30 times with the loop's other statements switched on one by one (GPU box: python tools/host_gap2.py)."""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch


def main():
    import bench
    from fedicra_amd import ops
    a = argparse.Namespace(batch=12, size=512, in_chns=3, classes=3, round_iters=10, loader_batches=8, data="host", no_graph=False,
                           rccl_single_rank=False)
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    fed = bench.Federation(a, 0, 1, dev, "bf16")
    fed.run_steps(30)
    torch.cuda.synchronize()
    c = fed.client
    rec = c._steps["head"]
    hist = torch.zeros((64, 5), dtype=torch.float32, device=dev)
    batches = c.sampled_batches or list(c.trainloader)

    def loop(stage, histw, freeze, prefetch, n=30, sync_each=False):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        th = 0.0
        with c._scope():
            for i in range(n):
                b = batches[i % len(batches)]
                if stage:
                    c._stage(b)
                if freeze:
                    c._set_freeze(0)
                rec.graph.replay()
                ops.bump_weights_epoch()
                if histw:
                    t1 = time.perf_counter()
                    hist[i, 0] = rec.loss
                    hist[i, 1] = rec.loss_ce
                    hist[i, 2] = rec.loss_lc
                    th += time.perf_counter() - t1
                if prefetch:
                    st = c._stager()
                    if st is not None:
                        st.prefetch(batches[(i + 1) % len(batches)])
        t_host = time.perf_counter() - t0
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        return el / n * 1e3, t_host / n * 1e3, th / n * 1e3

    for name, kw in [("replay only", dict(stage=False, histw=False, freeze=False, prefetch=False)),
                     ("+ stage", dict(stage=True, histw=False, freeze=False, prefetch=False)),
                     ("+ stage + prefetch", dict(stage=True, histw=False, freeze=False, prefetch=True)),
                     ("+ stage + prefetch + hist writes", dict(stage=True, histw=True, freeze=False, prefetch=True)),
                     ("+ stage + prefetch + hist + set_freeze (the loop)", dict(stage=True, histw=True, freeze=True, prefetch=True)),
                     ("hist writes only", dict(stage=False, histw=True, freeze=False, prefetch=False)),
                     ("replay only (again)", dict(stage=False, histw=False, freeze=False, prefetch=False))]:
        el, th, thist = loop(**kw)
        print(f"{name:52s}: {el:7.3f} ms per iteration wall, host enqueue {th:6.3f} ms (hist writes {thist:6.3f})")


if __name__ == "__main__":
    main()
