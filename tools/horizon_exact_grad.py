"""How far from the CPU fp32 reference is an implementation whose gradients are EXACT (fp64 backward, rounded to fp32),
after k = 1 AdamW step?  (the best case of an fp64-accumulating HIP parity mode)"""
import sys, copy
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import numpy as np, torch
from fedicra_amd.minifed import make_data
from oracle.unet_ref import RefUNet, seeded_state
from oracle.losses_ref import pce_loss

data, val = make_data()
b = data[0][0]
x, y = b["image"].unsqueeze(1), b["label"]

def grads(dtype, threads, mask_seed=0):
    torch.set_num_threads(threads)
    m = RefUNet(1, 2); seeded_state(m, 2022); m.train()
    if dtype == torch.float64: m = m.double()
    torch.manual_seed(mask_seed)
    out = m(x.to(dtype))
    loss = pce_loss(out[0], y, 2)
    loss.backward()
    return m, {n: p.grad.detach().clone() for n, p in m.named_parameters()}

def step_and_logits(g32, threads=8):
    torch.set_num_threads(threads)
    m = RefUNet(1, 2); seeded_state(m, 2022); m.train()
    opt = torch.optim.AdamW(m.parameters(), lr=0.01, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2)
    for n, p in m.named_parameters():
        p.grad = g32[n].float().clone()
    opt.step()
    torch.manual_seed(77)
    with torch.no_grad():
        return m(x)[0].numpy(), m

m8, g8 = grads(torch.float32, 8)
m1, g1 = grads(torch.float32, 1)
m64, g64 = grads(torch.float64, 8)
# NOTE: fp64 forward differs from fp32 forward too; its gradient is "exact" to ~1e-16
l8, n8 = step_and_logits(g8)
l1, n1 = step_and_logits(g1)
l64, n64 = step_and_logits({k: v.float() for k, v in g64.items()})
print("k=1 train-logit max |d|: cpu(1 thread grads) vs cpu(8): %.3e ; exact(fp64) grads vs cpu(8): %.3e ; scale %.2f" % (
    np.abs(l1 - l8).max(), np.abs(l64 - l8).max(), np.abs(l8).max()))
tot = flips1 = flips64 = 0
big = []
for n in g8:
    a, b1, c = g8[n].flatten(), g1[n].flatten(), g64[n].float().flatten()
    tot += a.numel()
    f1 = (torch.sign(a) != torch.sign(b1)); f64 = (torch.sign(a) != torch.sign(c))
    flips1 += int(f1.sum()); flips64 += int(f64.sum())
    if int(f64.sum()):
        rel = (a - c).abs().max() / (c.abs().max() + 1e-30)
        big.append((n, int(f64.sum()), a.numel(), float(c.abs().max()), float(c.abs()[f64].max()), float(rel)))
print(f"parameters {tot}: sign(cpu8) != sign(cpu1): {flips1}; sign(cpu8) != sign(exact): {flips64}")
for r in sorted(big, key=lambda r: -r[1])[:25]:
    print("  %-50s flips %6d / %7d  max|g| %.2e  max|g| among flipped %.2e  max rel err %.1e" % r)
# update distance in parameters
d64 = max(float((p - q).abs().max()) for p, q in zip(n64.parameters(), n8.parameters()))
d1 = max(float((p - q).abs().max()) for p, q in zip(n1.parameters(), n8.parameters()))
print("max parameter distance after the step: exact vs cpu8 %.3e, cpu1 vs cpu8 %.3e (2 lr = 0.02)" % (d64, d1))
