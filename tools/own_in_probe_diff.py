#!/usr/bin/env python
"""Head-phase iteration with the own forward as group 0 of the batched LC forwards (MyClient.own_in_probe) against the own forward
as a pass of its own: which entries of the state differ after N iterations, and by how much.  GPU box:
    python tools/own_in_probe_diff.py [--iters 1] [--dtype fp32]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=1)
    ap.add_argument("--rep", type=int, default=0)
    ap.add_argument("--dtype", default="fp32")
    a = ap.parse_args()
    from fedicra_amd import ops
    from fedicra_amd.flower_common import MyModel
    from fedicra_amd.flower_pCE_2D import MyClient
    from fedicra_amd.networks import net_factory
    from fedicra_amd.networks.unet import set_compute_dtype
    from helpers import loader
    res = []
    for merged in (False, True):
        args = argparse.Namespace(strategy="FedICRA", amp=0, model="unet_lc", cid=1, min_num_clients=4, num_classes=2, img_class="faz",
                                  base_lr=0.01, max_iterations=200, iters=a.iters, rep_iters=a.rep, alpha=1.0, snapshot_path=None,
                                  use_graph=False)
        torch.manual_seed(2022)
        ops.manual_seed(11)
        net = net_factory(args, net_type="unet_lc", in_chns=1, class_num=2).cuda()
        set_compute_dtype(net, a.dtype)
        batches = loader(3, 4, 64, cid=1, device="cuda")
        client = MyClient(args, MyModel(args, net, batches, batches), batches, batches)
        client.own_in_probe = merged
        client._train({"iter_global": 60, "iters": a.iters, "eval_iters": 10, "batch_size": 4, "stage": "fit"})
        torch.cuda.synchronize()
        sd = {k: v.detach().double().clone() for k, v in net.state_dict().items()}
        gr = {n: p._fi_gview.detach().double().clone() for n, p in net.named_parameters() if getattr(p, "_fi_gview", None) is not None}
        res.append((list(client.last_losses), sd, gr))
    (l0, s0, g0), (l1, s1, g1) = res
    print("losses", l0, l1)
    rows = []
    for k in s0:
        d = float((s0[k] - s1[k]).abs().max())
        rows.append((d / (float(s0[k].abs().max()) + 1e-12), d, k))
    rows.sort(reverse=True)
    for r in rows[:16]:
        print("state  rel %.3e abs %.3e  %s" % r)
    rows = []
    for k in g0:
        d = float((g0[k] - g1[k]).abs().max())
        rows.append((d / (float(g0[k].abs().max()) + 1e-12), d, k))
    rows.sort(reverse=True)
    for r in rows[:6]:
        print("grad   rel %.3e abs %.3e  %s" % r)


main()
