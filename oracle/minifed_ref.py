"""TEST INFRASTRUCTURE (checker side): the miniature federation of fedicra_amd/minifed.py run on the CPU restatement of the
reference -- `oracle.unet_ref.RefUNet` through `oracle.fed_ref.local_train` (/root/reference/code/flower_pCE_2D.py:51-181),
`fedavg_aggregate` (flwr 1.0.0 `aggregate`, restated) and `set_weights_plain` (code/flower_common.py:627-633), then the
validation Dice of `code/val_2D.py:9-22,25-74` (`oracle.losses_ref.eval_case`).  Same data objects, same dropout-mask
seeds (`seed + 100*round + cid`) as the HIP side.  Imported only by tests/ and by bench.py's cpu_baseline leg."""
from __future__ import annotations

import torch

from . import fed_ref
from .losses_ref import eval_case
from .unet_ref import RefUNet, seeded_state


def init_state(net, seed=2022):
    """The seeded initial state both sides start from (works on RefUNet and on fedicra_amd's UNet: same keys)."""
    seeded_state(net, seed)


def val_dice(ref, val):
    was = ref.training
    ref.eval()
    d = 0.0
    with torch.no_grad():
        for b in val:
            pred = ref(b["image"].unsqueeze(1))[0].argmax(1)[0].numpy()
            d += eval_case(pred, b["label"][0].numpy(), 2)[0]
    ref.train(was)
    return d / len(val)


def run_oracle(data, val, *, rounds=3, iters=8, n_k=(3, 2), seed=0, max_iterations=200, threads=None, perturb=0.0,
               trace=None):
    """-> dict(dice, losses, net).  `perturb`: relative Gaussian perturbation of the initial weights (the oracle's own
    sensitivity, for the spread the HIP path is judged against); `threads`: torch CPU threads for this run."""
    old_threads = torch.get_num_threads()
    if threads:
        torch.set_num_threads(int(threads))
    try:
        K = len(data)
        refs = [RefUNet(1, 2) for _ in range(K)]
        for r in refs:
            init_state(r)
            if perturb:
                g = torch.Generator().manual_seed(991)
                with torch.no_grad():
                    for p in r.parameters():
                        p.mul_(1.0 + perturb * torch.randn(p.shape, generator=g))
        states = [fed_ref.TrainState(0.01) for _ in range(K)]
        losses = []
        for rnd in range(rounds):
            res = []
            for cid in range(K):
                torch.manual_seed(seed + 100 * rnd + cid)
                _, met = fed_ref.local_train(refs[cid], states[cid], data[cid], iters=iters, num_classes=2, base_lr=0.01,
                                             max_iterations=max_iterations)
                res.append((fed_ref.get_weights(refs[cid]), n_k[cid]))
                losses.append(float(met["loss"][-1]))
            glob = fed_ref.fedavg_aggregate(res)
            for r in refs:
                fed_ref.set_weights_plain(r, glob)
            if trace is not None:
                trace(rnd, refs)
        return {"dice": float(val_dice(refs[0], val)), "losses": losses, "net": refs[0]}
    finally:
        torch.set_num_threads(old_threads)
