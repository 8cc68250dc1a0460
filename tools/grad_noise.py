#!/usr/bin/env python
"""Gradient round-off of the HIP fp32 parity mode against the CPU oracle's (GPU box; VERDICT r4 'k = 1 parity horizon').

One training iteration's backward pass of UNet(1, 2) on the miniature federation's first batch (4 x 1 x 64 x 64), identical
dropout masks, three ways: the oracle in fp64 (the exact gradient), the oracle in fp32 on 8 CPU threads (the reference's
arithmetic), the HIP path in fp32 mode.  Per parameter tensor: max |g|, the fp32 oracle's and the HIP path's max error
against fp64, and how many elements take the other SIGN than the fp32 oracle (AdamW's first step is lr * g / (|g| + 1e-8):
a sign flip of a gradient above ~1e-8 moves the weight by 2 lr).  Checker-side tool: imports oracle/."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch


def main():
    from fedicra_amd import ops
    from fedicra_amd.minifed import make_data
    from fedicra_amd.networks.unet import UNet, set_compute_dtype
    from fedicra_amd.optim import FusedAdamW
    from oracle.losses_ref import pce_loss
    from oracle.unet_ref import RefUNet, seeded_state
    data, _ = make_data()
    b = data[0][0]
    x, y = b["image"].unsqueeze(1), b["label"]
    seed = int(os.environ.get("SEED", "0"))

    def oracle(dtype, threads):
        torch.set_num_threads(threads)
        m = RefUNet(1, 2)
        seeded_state(m, 2022)
        m.train()
        if dtype == torch.float64:
            m = m.double()
        torch.manual_seed(seed)
        out = m(x.to(dtype))
        pce_loss(out[0], y, 2).backward()
        return {n: p.grad.detach().double().clone() for n, p in m.named_parameters()}, out[0].detach().double()

    g64, l64 = oracle(torch.float64, 8)
    g32, l32 = oracle(torch.float32, 8)
    g31, _ = oracle(torch.float32, 1)
    net = UNet(1, 2)
    seeded_state(net, 2022)
    net = net.cuda().train()
    set_compute_dtype(net, "fp32")
    ops.set_dropout_mask_provider(lambda shape, p: torch.empty(shape).bernoulli_(1 - p))
    try:
        torch.manual_seed(seed)
        opt = FusedAdamW(net, lr=0.01, base_lr=0.01, max_iterations=30000)
        opt.zero_grad()
        ops.begin_iteration(torch.device("cuda"))
        out = net(x.cuda())
        loss = ops.ce_loss(out[0].permute(0, 2, 3, 1), y.cuda(), 2)
        loss.backward()
        ops.flush_wgrad()
    finally:
        ops.set_dropout_mask_provider(None)
    torch.cuda.synchronize()
    gh = {n: p.grad.detach().double().cpu() for n, p in net.named_parameters()}
    lh = out[0].detach().double().cpu()
    print(f"forward logits max |d| vs fp64 oracle: cpu32 {float((l32 - l64).abs().max()):.2e}  hip32 {float((lh - l64).abs().max()):.2e}")
    print(f"{'parameter':52s} {'numel':>8s} {'max|g|':>9s} {'err cpu32':>9s} {'err cpu1t':>9s} {'err hip':>9s} {'hip/cpu':>7s} "
          f"{'flip c1':>7s} {'flip hip':>8s} {'flip>1e-8':>9s}")
    tot = [0, 0, 0]
    rows = []
    for n in g64:
        e, c, c1, h = g64[n].flatten(), g32[n].flatten(), g31[n].flatten(), gh[n].flatten()
        ec, e1, eh = float((c - e).abs().max()), float((c1 - e).abs().max()), float((h - e).abs().max())
        f1 = int((torch.sign(c) != torch.sign(c1)).sum())
        fh = torch.sign(c) != torch.sign(h)
        fbig = int((fh & (c.abs() > 1e-8)).sum())
        tot[0] += f1
        tot[1] += int(fh.sum())
        tot[2] += fbig
        rows.append((n, e.numel(), float(e.abs().max()), ec, e1, eh, eh / max(ec, 1e-30), f1, int(fh.sum()), fbig))
    for r in rows:
        print("%-52s %8d %9.2e %9.2e %9.2e %9.2e %7.1f %7d %8d %9d" % r)
    print(f"sign flips against the fp32 oracle (8 threads): oracle on 1 thread {tot[0]}, HIP fp32 {tot[1]} "
          f"(of which the oracle's |g| > 1e-8, i.e. a 2 lr move: {tot[2]})")


if __name__ == "__main__":
    main()
