#!/usr/bin/env python
"""Where in the backward pass does the fp32 parity mode leave the fp64 gradient?  (GPU box.)  Every BatchNorm-backward launch of
one training iteration (tools/grad_noise.py's setting) with its incoming gradient dz and its outgoing gradient dy, matched to
the oracle layer of the same shape whose fp64 tensors are closest; the fp32 CPU oracle's own errors beside them."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn as nn


def main():
    from fedicra_amd import _lib as L
    from fedicra_amd import ops
    from fedicra_amd.minifed import make_data
    from fedicra_amd.networks.unet import UNet, set_compute_dtype
    from fedicra_amd.optim import FusedAdamW
    from oracle.losses_ref import pce_loss
    from oracle.unet_ref import RefUNet, seeded_state
    data, _ = make_data()
    b = data[0][0]
    x, y = b["image"].unsqueeze(1), b["label"]
    seed = 0

    def oracle(dtype):
        torch.set_num_threads(8)
        m = RefUNet(1, 2)
        seeded_state(m, 2022)
        m.train()
        if dtype == torch.float64:
            m = m.double()
        recs = {}
        hooks = []
        for name, mod in m.named_modules():
            if isinstance(mod, nn.Sequential) and len(mod) == 7 and isinstance(mod[1], nn.BatchNorm2d):
                for idx, tag in ((0, "dy1"), (3, "dz1"), (4, "dy2"), (6, "dz2")):
                    def fh(mod_, inp, out, key=(name, tag)):
                        out.register_hook(lambda g, key=key: recs.__setitem__(key, g.detach().double().clone()))
                    hooks.append(mod[idx].register_forward_hook(fh))
        torch.manual_seed(seed)
        out = m(x.to(dtype))
        pce_loss(out[0], y, 2).backward()
        for h in hooks:
            h.remove()
        return recs

    r64, r32 = oracle(torch.float64), oracle(torch.float32)
    net = UNet(1, 2)
    seeded_state(net, 2022)
    net = net.cuda().train()
    set_compute_dtype(net, "fp32")
    calls = []
    orig = L.bn_act_bwd_apply

    def spy(*a, **k):
        r = orig(*a, **k)
        dz, dy = a[0], a[8]
        calls.append((dz.detach().clone(), None if dy is None else dy.detach().clone()))
        return r

    L.bn_act_bwd_apply = spy
    ops.set_dropout_mask_provider(lambda shape, p: torch.empty(shape).bernoulli_(1 - p))
    try:
        torch.manual_seed(seed)
        opt = FusedAdamW(net, lr=0.01, base_lr=0.01, max_iterations=30000)
        opt.zero_grad()
        ops.begin_iteration(torch.device("cuda"))
        out = net(x.cuda())
        ops.ce_loss(out[0].permute(0, 2, 3, 1), y.cuda(), 2).backward()
        ops.flush_wgrad()
    finally:
        ops.set_dropout_mask_provider(None)
        L.bn_act_bwd_apply = orig
    torch.cuda.synchronize()

    def rel(a, ref):
        return float((a - ref).abs().max() / (ref.abs().max() + 1e-300))

    layers = sorted({k[0] for k in r64})
    print(f"{'#':>2s} {'matched oracle layer':44s} {'shape':>18s} {'max|dz|':>9s} {'dz hip':>9s} {'dz cpu32':>9s} {'dy hip':>9s} {'dy cpu32':>9s} "
          f"{'sum(dz) hip':>11s} {'cpu32':>9s}")
    for i, (dz, dy) in enumerate(calls):
        dzc = dz.double().cpu().permute(0, 3, 1, 2)
        best = None
        for name in layers:
            for half in ("1", "2"):
                ref = r64.get((name, "dz" + half))
                if ref is None or tuple(ref.shape) != tuple(dzc.shape):
                    continue
                e = rel(dzc, ref)
                if best is None or e < best[0]:
                    best = (e, name, half)
        if best is None:
            print(f"{i:2d} no oracle layer of shape {tuple(dzc.shape)}")
            continue
        e, name, half = best
        ref_dz, ref_dy = r64[(name, "dz" + half)], r64[(name, "dy" + half)]
        c_dz, c_dy = r32[(name, "dz" + half)], r32[(name, "dy" + half)]
        dyc = dy.double().cpu().permute(0, 3, 1, 2) if dy is not None else None
        s_ref = ref_dz.sum((0, 2, 3))
        s_h = rel(dzc.sum((0, 2, 3)), s_ref)
        s_c = rel(c_dz.sum((0, 2, 3)), s_ref)
        print(f"{i:2d} {name + ' half ' + half:44s} {str(tuple(dzc.shape)):>18s} {float(ref_dz.abs().max()):9.2e} {e:9.2e} {rel(c_dz, ref_dz):9.2e} "
              f"{(rel(dyc, ref_dy) if dyc is not None else float('nan')):9.2e} {rel(c_dy, ref_dy):9.2e} {s_h:11.2e} {s_c:9.2e}")


if __name__ == "__main__":
    main()
