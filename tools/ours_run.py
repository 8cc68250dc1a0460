"""The README's training command on one GPU client (functional + timing): procedure flower_pCE_2D_GateCRFMsacleTreeEnergyLoss_Ours,
--model unet_lc_multihead --strategy FedICRA --alpha 1 --rep_iters 3, FAZ-like 12x1x256x256 batches, 5 clients
(4 extra no-grad forwards per iteration for the LC loss), pCE + tree-energy + 0.1 gated-CRF [+ LC], hipGraph per phase.
    python tools/ours_run.py [--dtype bf16|fp32] [--size 256] [--clients 5]"""
import argparse
import sys
import time

import torch

sys.path.insert(0, '.')
ap = argparse.ArgumentParser()
ap.add_argument("--dtype", default="bf16")
ap.add_argument("--size", type=int, default=256)
ap.add_argument("--clients", type=int, default=5)
ap.add_argument("--iters", type=int, default=10)
ap.add_argument("--profile", action="store_true")
a = ap.parse_args()
from fedicra_amd.flower_common import MyModel
from fedicra_amd.flower_pCE_2D_GateCRFMsacleTreeEnergyLoss_Ours import MyClient
from fedicra_amd.networks import net_factory
from fedicra_amd.networks.unet import set_compute_dtype
from fedicra_amd.synth import phantom_batch
dev = torch.device("cuda", 0)
args = argparse.Namespace(strategy="FedICRA", amp=0, model="unet_lc_multihead", cid=2, min_num_clients=a.clients, num_classes=2,
                          img_class="faz", base_lr=0.01, max_iterations=30000, iters=a.iters, rep_iters=3, alpha=1.0,
                          snapshot_path=None, use_graph=True, tree_loss_weight=0.1)
torch.manual_seed(2022)
net = net_factory(args, net_type="unet_lc_multihead", in_chns=1, class_num=2).to(dev)
set_compute_dtype(net, a.dtype)
batches = []
for i in range(3):
    img, weak, _ = phantom_batch(12, a.size, 1, 2, cid=2, index=i, labeled_frac=0.05)
    batches.append({"image": torch.from_numpy(img).to(dev), "label": torch.from_numpy(weak).to(dev)})
model = MyModel(args, net, batches, batches)
client = MyClient(args, model, batches, batches)
for r in range(4):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    loss, m = client._train({"iter_global": 60 + r, "iters": a.iters, "eval_iters": 99, "batch_size": 12, "stage": "fit"})
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"round {r}: loss {loss:.4f}  {dt * 1e3 / a.iters:.2f} ms/iter  {12 * a.iters / dt:.1f} images/s  "
          f"losses {[round(v, 4) for v in client.last_losses]}")
print("peak GB", torch.cuda.max_memory_allocated() / 1e9)
if "--profile" in sys.argv or True:
    from fedicra_amd import _lib as L
    client.use_graph = False
    cfg = {"iter_global": 70, "iters": 3, "eval_iters": 99, "batch_size": 12, "stage": "fit"}
    args.iters = 3
    client._train(cfg)
    L.profile_begin()
    client._train(cfg)
    kp = L.profile_end()
    prof = kp.summary()
    fam = {}
    for k, v in prof.items():
        f = fam.setdefault(k[0], [0, 0.0])
        f[0] += v["calls"]
        f[1] += v["ms"]
    tot = sum(v[1] for v in fam.values())
    print(f"C-ABI launches, eager, per iteration (sum {tot / 3:.2f} ms; torch glue kernels not included):")
    for k, v in sorted(fam.items(), key=lambda kv: -kv[1][1])[:14]:
        print(f"  {k:24s} {v[0] / 3:6.1f} launches  {v[1] / 3:7.3f} ms")
    top = sorted(prof.items(), key=lambda kv: -kv[1]["ms"])[:8]
    for k, v in top:
        print("   ", k, f"{v['calls'] / 3:.1f} x {v['ms'] / v['calls'] * 1e3:.1f} us")
