#!/usr/bin/env python
"""Does a pinned host -> device copy on a side stream overlap with compute on the main stream?  Times a compute-only loop, a
copy-only loop and both together (a conv kernel chain as the compute, 40 MB copies like bench.py's batches)."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fedicra_amd import _lib as L  # noqa: E402

dev = "cuda"
x = torch.randn(84, 128, 128, 64, device=dev).to(torch.bfloat16)
w = (torch.randn(64, 3, 3, 64, device=dev) * 0.05).to(torch.bfloat16)
y = torch.empty(84, 128, 128, 64, device=dev, dtype=torch.bfloat16)
host = torch.randn(12, 3, 512, 512).pin_memory()
dst = torch.empty_like(host, device=dev)
side = torch.cuda.Stream()


def compute(n=20):
    for _ in range(n):
        L.conv2d_fwd(x, None, w, None, y, None, None, ksize=3)


g = torch.cuda.CUDAGraph()
compute(2)
torch.cuda.synchronize()
with torch.cuda.graph(g):
    compute(20)


def timed(fn, reps=10):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


def copy_only():
    with torch.cuda.stream(side):
        dst.copy_(host, non_blocking=True)


def both():
    with torch.cuda.stream(side):
        dst.copy_(host, non_blocking=True)
    g.replay()


print(f"compute (graph of 20 convs) {timed(g.replay):.3f} ms | copy 40 MB H2D {timed(copy_only):.3f} ms | both {timed(both):.3f} ms")
# the same copy as a kernel that reads the pinned buffer directly (zero-copy over PCIe): torch elementwise copy from a
# device-mapped view is not exposed, so time the SDMA path with several chunks instead
chunks = host.chunk(8)
dchunks = dst.chunk(8)


def both_chunked():
    with torch.cuda.stream(side):
        for a, b in zip(dchunks, chunks):
            a.copy_(b, non_blocking=True)
    g.replay()


print(f"both, copy in 8 chunks {timed(both_chunked):.3f} ms; GPU_MAX_HW_QUEUES={os.environ.get('GPU_MAX_HW_QUEUES')}")

# ---- the bench's pattern: BatchStager prefetch of batch i+1 right after enqueueing step i
from fedicra_amd.staging import BatchStager  # noqa: E402
batches = [{"image": torch.randn(12, 3, 512, 512).pin_memory(), "label": torch.zeros(12, 512, 512, dtype=torch.uint8).pin_memory()}
           for _ in range(4)]
xbuf = torch.empty(12, 3, 512, 512, device=dev)
ybuf = torch.empty(12, 512, 512, dtype=torch.uint8, device=dev)


def run(steps, prefetch, two_slots=False):
    st = BatchStager(torch.device(dev))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        b = batches[i % 4]
        xs, ys = st.fetch(b)
        xbuf.copy_(xs, non_blocking=True)
        ybuf.copy_(ys, non_blocking=True)
        st.release()
        g.replay()
        if prefetch:
            st.prefetch(batches[(i + 1) % 4])
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


run(4, True)
print(f"stager loop: no prefetch {run(20, False):.3f} ms/step | prefetch {run(20, True):.3f} ms/step | compute alone {timed(g.replay):.3f}")

