"""A miniature federation on phantom data, run end to end on the HIP path: the third leg of BASELINE.json's metric
("Dice vs CPU ref"; /root/reference/code/val_2D.py:9-74, code/flower_common.py:122-136) needs a trained model, so this is
what `bench.py` trains -- outside its timed region -- in fp32 and bf16, and what the parity-horizon tests train, while the
CPU restatement of the reference (oracle/minifed_ref.py: the checker, never imported from here) goes through the SAME
rounds on the SAME data with the SAME dropout masks.

K FedAvg clients x `rounds` rounds x `iters` local iterations (flower_pCE_2D.py:51-181: fresh AdamW per round, poly LR),
weighted aggregation with n_k = #batches (flower_common.py:72), the global state loaded into every client (:627-633),
then `evaluate` of client 0 on a dense-mask validation set (:122-136).  Dropout masks come from the host generator in the
reference's draw order (`ops.set_dropout_mask_provider`), seeded `seed + 100*round + cid` per local-training call."""
from __future__ import annotations

import argparse

import torch

from .synth import phantom_batch


def make_data(K=2, n_k=(3, 2), batch=4, size=64, n_val=16, labeled_frac=0.3):
    """(per-client training batch lists, validation cases) as host tensors; the oracle side is fed the same objects."""
    data = []
    for cid in range(K):
        bs = []
        for i in range(n_k[cid]):
            img, weak, _ = phantom_batch(batch, size, 1, 2, cid=cid, index=i, labeled_frac=labeled_frac)
            bs.append({"image": torch.from_numpy(img), "label": torch.from_numpy(weak)})
        data.append(bs)
    vimg, _, vmask = phantom_batch(n_val, size, 1, 2, cid=7, dense=True)
    val = [{"image": torch.from_numpy(vimg[i:i + 1]), "label": torch.from_numpy(vmask[i:i + 1])} for i in range(n_val)]
    return data, val


def run_hip(data, val, *, dtype="fp32", rounds=3, iters=8, n_k=(3, 2), seed=0, init_state=None, max_iterations=200,
            device="cuda", trace=None):
    """-> dict(dice=val_mean_dice of client 0 after the last round, losses=[last loss per (round, client)], net=client 0's
    model).  `init_state`: a callable(net) that loads the seeded initial state both sides start from.  `trace(round, nets)`:
    called after every round's global load (the parity-horizon tests read logits there)."""
    from . import ops
    from .flower_common import MyModel, aggregate_device, evaluate
    from .flower_pCE_2D import MyClient
    from .networks.unet import UNet, set_compute_dtype
    K = len(data)
    clients = []
    for cid in range(K):
        args = argparse.Namespace(strategy="FedAvg", amp=0, model="unet", cid=cid, min_num_clients=K, num_classes=2,
                                  img_class="faz", base_lr=0.01, max_iterations=max_iterations, iters=iters, rep_iters=3,
                                  alpha=0.5, snapshot_path=None, use_graph=False)
        net = UNet(1, 2)
        if init_state is not None:
            init_state(net)
        net = net.to(device)
        set_compute_dtype(net, dtype)
        m = MyModel(args, net, data[cid], data[cid])
        m.verbose = False
        clients.append(MyClient(args, m, data[cid], data[cid]))
    ops.set_dropout_mask_provider(lambda shape, p: torch.empty(shape).bernoulli_(1 - p))
    losses = []
    try:
        for rnd in range(rounds):
            res = []
            for cid in range(K):
                torch.manual_seed(seed + 100 * rnd + cid)
                last, _ = clients[cid]._train({"iter_global": rnd, "iters": iters, "eval_iters": 99, "batch_size": 4,
                                               "stage": "fit"})
                losses.append(float(last))
                res.append((clients[cid].model.get_device_weights(), n_k[cid]))
            glob = aggregate_device(res)
            for c in clients:
                c.model.set_weights(glob, {"iter_global": rnd})
            if trace is not None:
                trace(rnd, [c._net() for c in clients])
    finally:
        ops.set_dropout_mask_provider(None)
    met = evaluate(clients[0].args, clients[0].model.model, val)
    return {"dice": float(met["val_mean_dice"]), "losses": losses, "net": clients[0]._net(), "metrics": met}
