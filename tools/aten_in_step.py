#!/usr/bin/env python
"""Which launches of the c3 training step / ALA batch are NOT this library's?  torch's profiler over eager iterations of the
bench's own federation (no hipGraph), every ATen operator that launched a device kernel with its input shapes, call count and
device time -- per head-phase iteration, body-phase iteration and ALA epoch.  GPU box:  python tools/aten_in_step.py [--c4]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from torch.profiler import ProfilerActivity, profile


def table(prof, title, per):
    rows = []
    for e in prof.key_averages(group_by_input_shape=True):
        dev = getattr(e, "self_device_time_total", 0) or getattr(e, "self_cuda_time_total", 0)
        if dev <= 0 or not e.key.startswith("aten::"):
            continue
        rows.append((dev / per, e.count / per, e.key, str(e.input_shapes)[:110]))
    rows.sort(reverse=True)
    print(f"--- {title}: ATen operators with device time, per {per} unit(s); total {sum(r[0] for r in rows):.1f} us")
    for us, cnt, key, shp in rows[:25]:
        print(f"{us:9.1f} us {cnt:6.1f} x  {key:28s} {shp}")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--c4", action="store_true")
    a0 = ap.parse_args()
    import bench
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    if a0.c4:
        a = argparse.Namespace(no_graph=True, batch=2, size=128, round_iters=10)
        vol = bench.Volumes(a, 0, 1, dev, "bf16", kind="c4")
        for _ in range(4):
            vol.step()
        torch.cuda.synchronize()
        with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
            for _ in range(4):
                vol.step()
            torch.cuda.synchronize()
        table(prof, "c4 iteration", 4)
        return
    a = argparse.Namespace(batch=12, size=512, in_chns=3, classes=3, round_iters=10, loader_batches=8, data="host", no_graph=True,
                           rccl_single_rank=False)
    fed = bench.Federation(a, 0, 1, dev, "bf16")
    fed.run_steps(20)
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
        fed.run_steps(10)
        torch.cuda.synchronize()
    table(prof, "one round (7 head-phase + 3 body-phase iterations + aggregation + ALA epoch)", 1)


main()
