#!/usr/bin/env python
"""Diagnostic: per-parameter gradient differences of one UNet step under subsets of the narrow forms (fi_narrow_tuning mask)
against the general kernels and against the fp32 compute mode."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fedicra_amd import ops, _lib as L
from fedicra_amd.networks.unet import UNet, set_compute_dtype
from oracle.unet_ref import seeded_state

DEV = "cuda"
torch.manual_seed(0)
x = torch.randn(8, 3, 128, 128, device=DEV)
y = torch.randint(0, 3, (8, 128, 128), device=DEV, dtype=torch.uint8)


def run(dtype, mask, scale):
    L.lib().fi_narrow_tuning(mask)
    m = UNet(3, 3)
    seeded_state(m, 5)
    m = m.cuda()
    set_compute_dtype(m, dtype)
    m.eval()
    out = m(x)[0]
    loss = ops.ce_loss(out.permute(0, 2, 3, 1), y, 3)
    (loss * scale).backward()
    ops.flush_wgrad()
    torch.cuda.synchronize()
    L.lib().fi_narrow_tuning(-1)
    return out.detach().float(), {k: p.grad.detach().float().clone() / scale for k, p in m.named_parameters() if p.grad is not None}


o32, g32 = run("fp32", 0, 1.0)
keys = ["encoder.in_conv.conv_conv.0.weight", "encoder.in_conv.conv_conv.0.bias", "encoder.in_conv.conv_conv.4.weight",
        "encoder.down4.maxpool_conv.1.conv_conv.4.weight", "decoder.up4.conv.conv_conv.4.weight", "decoder.out_conv.weight", "decoder.out_conv.bias"]
for dtype, scale in (("bf16", 1.0), ("fp16", 4096.0)):
    base_o, base = run(dtype, 0, scale)
    base2_o, base2 = run(dtype, 0, scale)
    print(dtype, "general vs general (determinism):", max((base[k] - base2[k]).abs().max().item() for k in base))
    for mask in (1, 2, 4, 7):
        o, g = run(dtype, mask, scale)
        print(f"{dtype} mask {mask}: logits diff vs general {(o - base_o).abs().max().item():.3e} (vs fp32: {(o - o32).abs().max().item():.3e}, general vs fp32 {(base_o - o32).abs().max().item():.3e})")
        for k in keys:
            s = g32[k].abs().max().item()
            print(f"    {k:55s} |narrow-general| {(g[k] - base[k]).abs().max().item() / s:.3e}   |narrow-fp32| {(g[k] - g32[k]).abs().max().item() / s:.3e}"
                  f"   |general-fp32| {(base[k] - g32[k]).abs().max().item() / s:.3e}")
