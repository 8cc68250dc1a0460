#!/usr/bin/env python
"""Probe: two co-located clients on streams restricted to disjoint halves of the CUs (hipExtStreamCreateWithCUMask)
versus unrestricted streams.  Timing only (shared arena: numerics not meaningful)."""
import argparse
import ctypes as C
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

hip = C.CDLL("libamdhip64.so")


def masked_stream(bits):
    n = 8                                   # 256 CUs -> 8 uint32 words
    arr = (C.c_uint32 * n)()
    for b in bits:
        arr[b // 32] |= 1 << (b % 32)
    s = C.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(C.byref(s), C.c_uint32(n), arr)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(s.value)


def main():
    a = argparse.Namespace(dtype="bf16", size=256, batch=12, round_iters=10, no_graph=False)
    dev = torch.device("cuda", 0)
    from fedicra_amd.flower_common import MyModel
    from fedicra_amd.flower_pCE_2D import MyClient
    from fedicra_amd.networks import net_factory
    from fedicra_amd.networks.unet import set_compute_dtype
    clients = []
    for k in range(2):
        args = bench.make_args(a, k, 2)
        torch.manual_seed(2022)
        net = net_factory(args, net_type="unet", in_chns=1, class_num=2)
        set_compute_dtype(net, a.dtype)
        loader = bench.device_loader(4, a.batch, a.size, k, dev)
        c = MyClient(args, MyModel(args, net, loader, loader), loader, loader)
        args.iters = 4
        c._train({"iter_global": 0, "iters": 4, "eval_iters": 40, "batch_size": a.batch, "stage": "fit"})
        torch.cuda.synchronize()
        clients.append(c)
    graphs = [c._steps["all"].graph for c in clients]
    configs = {
        "unrestricted": [torch.cuda.Stream(), torch.cuda.Stream()],
        "halves contiguous (0-127 | 128-255)": [masked_stream(range(0, 128)), masked_stream(range(128, 256))],
        "halves interleaved (even | odd)": [masked_stream(range(0, 256, 2)), masked_stream(range(1, 256, 2))],
        "3/4 overlapping (0-191 | 64-255)": [masked_stream(range(0, 192)), masked_stream(range(64, 256))],
    }
    reps = 200
    for name, streams in configs.items():
        for K in (1, 2):
            for _ in range(2):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for r in range(reps):
                    for g, s in zip(graphs[:K], streams[:K]):
                        with torch.cuda.stream(s):
                            g.replay()
                torch.cuda.synchronize()
                dt = time.perf_counter() - t0
            print(f"{name:40s} K={K}: {dt / reps * 1e3:.3f} ms per round, {K * a.batch * reps / dt:.0f} images/s")


main()
