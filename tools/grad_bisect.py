#!/usr/bin/env python
"""Which op of the fp32 parity mode loses gradient precision?  (tools/grad_noise.py: from the last ConvBlock's first half
backwards every HIP gradient is ~1000x further from an fp64 backward than the CPU's fp32 one.)  Each autograd op of the
training path alone, fp32, against the same op in torch on the CPU in fp64 -- with torch CPU fp32 as the yardstick:
max |error| / max |reference| of every output and gradient."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn as nn
import torch.nn.functional as F


def rel(a, ref):
    return float((a.double().cpu() - ref).abs().max() / (ref.abs().max() + 1e-300))


def main():
    from fedicra_amd import ops
    dev = "cuda"
    torch.manual_seed(5)

    def conv_block(N, H, W, c0, c1, cout, k, bn_on):
        cin = c0 + c1
        conv = nn.Conv2d(cin, cout, k, padding=k // 2)
        bn = nn.BatchNorm2d(cout)
        with torch.no_grad():
            bn.weight.uniform_(0.5, 1.5), bn.bias.uniform_(-0.5, 0.5)
        x = torch.randn(N, cin, H, W)
        gz = torch.randn(N, cout, H, W) * (1.0 + torch.arange(cout).view(1, -1, 1, 1) * 0.1)
        out = {}
        for name, dt in (("f64", torch.float64), ("cpu32", torch.float32)):
            c, b = nn.Conv2d(cin, cout, k, padding=k // 2).to(dt), nn.BatchNorm2d(cout).to(dt)
            c.load_state_dict({kk: v.to(dt) for kk, v in conv.state_dict().items()})
            b.load_state_dict({kk: (v.to(dt) if v.is_floating_point() else v) for kk, v in bn.state_dict().items()})
            b.train()
            xi = x.to(dt).clone().requires_grad_(True)
            y = c(xi)
            z = F.leaky_relu(b(y), 0.01) if bn_on else y
            z.backward(gz.to(dt))
            out[name] = {"z": z.detach().double(), "dx": xi.grad.double(), "dw": c.weight.grad.double(), "db": c.bias.grad.double()}
            if bn_on:
                out[name].update({"dgamma": b.weight.grad.double(), "dbeta": b.bias.grad.double()})
        cg, bg = conv.to(dev), bn.to(dev).train()
        xh = x.permute(0, 2, 3, 1).contiguous().to(dev)
        x0 = xh[..., :c0].contiguous().requires_grad_(True)
        x1 = xh[..., c0:].contiguous().requires_grad_(True) if c1 else None
        ops.begin_iteration(torch.device(dev))
        z = ops.conv_bn_act(x0, x1, cg, bg, 0.01) if bn_on else ops.conv2d(x0, x1, cg)
        z.backward(gz.permute(0, 2, 3, 1).contiguous().to(dev))
        ops.flush_wgrad()
        torch.cuda.synchronize()
        dx = torch.cat([x0.grad] + ([x1.grad] if c1 else []), dim=3).permute(0, 3, 1, 2)
        hip = {"z": z.detach().permute(0, 3, 1, 2), "dx": dx, "dw": cg.weight.grad, "db": cg.bias.grad}
        if bn_on:
            hip.update({"dgamma": bg.weight.grad, "dbeta": bg.bias.grad})
        keys = list(out["f64"])
        print(f"{'conv+BN+LeakyReLU' if bn_on else 'conv':18s} {N}x{H}x{W} {c0}+{c1}->{cout} k{k}: " +
              "  ".join(f"{kk} hip {rel(hip[kk], out['f64'][kk]):.1e} / cpu {rel(out['cpu32'][kk], out['f64'][kk]):.1e}" for kk in keys))

    for bn_on in (False, True):
        conv_block(4, 64, 64, 16, 0, 16, 3, bn_on)
        conv_block(4, 64, 64, 16, 16, 16, 3, bn_on)
        conv_block(4, 16, 16, 64, 0, 64, 3, bn_on)
        conv_block(4, 4, 4, 256, 0, 256, 3, bn_on)
    conv_block(4, 64, 64, 16, 0, 2, 3, False)
    conv_block(4, 32, 32, 32, 0, 16, 1, False)

    # pooling / up-sampling
    for what in ("maxpool", "upsample"):
        x = torch.randn(4, 16, 32, 32)
        ref = {}
        for name, dt in (("f64", torch.float64), ("cpu32", torch.float32)):
            xi = x.to(dt).clone().requires_grad_(True)
            y = F.max_pool2d(xi, 2) if what == "maxpool" else F.interpolate(xi, scale_factor=2, mode="bilinear", align_corners=True)
            g = torch.randn(y.shape, generator=torch.Generator().manual_seed(1)).to(dt)
            y.backward(g)
            ref[name] = (y.detach().double(), xi.grad.double())
        xh = x.permute(0, 2, 3, 1).contiguous().to(dev).requires_grad_(True)
        y = ops.maxpool2(xh) if what == "maxpool" else ops.upsample2x(xh)
        g = torch.randn(ref["f64"][0].shape, generator=torch.Generator().manual_seed(1)).permute(0, 2, 3, 1).contiguous().to(dev)
        y.backward(g)
        print(f"{what:18s}: y hip {rel(y.detach().permute(0, 3, 1, 2), ref['f64'][0]):.1e} / cpu {rel(ref['cpu32'][0], ref['f64'][0]):.1e}  "
              f"dx hip {rel(xh.grad.permute(0, 3, 1, 2), ref['f64'][1]):.1e} / cpu {rel(ref['cpu32'][1], ref['f64'][1]):.1e}")


if __name__ == "__main__":
    main()
