#!/usr/bin/env python
"""Rank-per-GPU federated round driver: what ``flower_runner.py`` + the Flower server loop do for the hot path, on one
8 x MI355X node without a server process.

    python -m torch.distributed.run --nnodes=1 --nproc-per-node K --master-addr 127.0.0.1 --master-port P \\
        -m fedicra_amd.run_federated --procedure flower_pCE_2D --exp NAME --strategy FedICRA --model unet_lc \\
        --img_class faz --iters 10 --eval_iters 20 --max_iterations 30000 [--root_path ../data/FAZ_h5 | --synthetic 48]

The reference launches one server and K client OS processes that talk gRPC (/root/reference/code/flower_runner.py:96-121)
and pins client k to GPU k (``--gpus``, :100-102).  Here rank k IS client k on GPU k; the server's role is spread over the
ranks:

  * ``MyServer.fit``'s round loop (flower_common.py:258-365) runs identically on every rank: rounds
    ``iters, 2*iters, ...``, ``fit`` every round, ``evaluate`` when ``round % eval_iters == 0``, the same config dicts
    (flower_pCE_2D.py:320-338);
  * ``Server.fit_round`` -> ``FedAvg.aggregate_fit`` -> ``aggregate`` is ONE weighted all-reduce of the flat state over
    xGMI (comm.WeightedAllReduce, weights = ``FitRes.num_examples`` = #batches, quirk 5), after which every rank holds
    the global parameters -- FedICRA's clients then mix them in locally (``MyModel.set_weights``);
  * metric folds (``fit_metrics_aggregation_fn`` / ``evaluate_metrics_aggregation_fn``, :393-428) run on gathered scalars;
  * rank 0 writes the server's artefacts with the reference's names (:343-365, consumed by code/test.py:263-265):
    ``iter_{n}_dice_{d}.pth`` / ``{model}_best_model.pth`` (centralized strategies), ``client_{id}_iter_{n}_dice_{d}.pth`` /
    ``client_{id}_{model}_best_model.pth`` and the 3000-iteration snapshots.  As in the reference the per-client files hold
    the GLOBAL parameters: ``get_client_state_dict`` overwrites every key that is not in its (empty) ``local_keys`` (:244-254).

Initial parameters: the reference takes them from one randomly chosen client (flwr ``_get_initial_parameters``, :209);
every client process seeds with 2022 and builds the same model, so rank 0's state is broadcast.  TensorBoard scalars are
written when ``tensorboardX`` is importable (it is not in the build image); the gRPC transport, the server-side
``evaluate_fn`` process and the preview-image grid are out of scope.
"""
from __future__ import annotations

import argparse
import importlib
import os
import random
import timeit
from collections import OrderedDict

import numpy as np
import torch

PROCEDURES = ["flower_pCE_2D", "flower_pCE_2D_GateCRFMsacleTreeEnergyLoss_Ours"]       # flower_runner.py:58-60
# flower_runner.py:62-94: data root, classes, input channels and each client's supervision type
DATASETS = {
    "faz": ("../data/FAZ_h5", 2, 1, ["scribble_noisy", "keypoint", "block", "box", "scribble"]),
    "odoc": ("../data/ODOC_h5", 3, 3, ["scribble", "scribble_noisy", "scribble_noisy", "keypoint", "block"]),
    "polyp": ("../data/Polypdata_h5", 2, 3, ["keypoint", "scribble", "box", "block"]),
}
TRAIN_SCALAR_METRICS = ["lr", "total_loss", "loss_ce"]                                   # flower_pCE_2D.py:372


def build_parser():
    """The launcher's flags (flower_runner.py:18-55) plus the data / output locations the client script derives."""
    p = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    p.add_argument("--procedure", type=str, default="flower_pCE_2D", choices=PROCEDURES)
    p.add_argument("--exp", type=str, required=True, help="experiment_name")
    p.add_argument("--base_lr", type=float, default=0.01)
    p.add_argument("--model", type=str, default="unet")
    p.add_argument("--img_class", type=str, default="faz", choices=sorted(DATASETS))
    p.add_argument("--max_iterations", type=int, default=30000)
    p.add_argument("--iters", type=int, default=10)
    p.add_argument("--eval_iters", type=int, default=20)
    p.add_argument("--alpha", type=float, default=0.5)
    p.add_argument("--batch_size", type=int, default=12)
    p.add_argument("--tree_loss_weight", type=float, default=0.1)
    p.add_argument("--strategy", type=str, default="FedAvg")
    p.add_argument("--img_size", type=int, default=256)
    p.add_argument("--amp", type=int, default=0)
    p.add_argument("--rep_iters", type=int, default=3)
    p.add_argument("--seed", type=int, default=2022)
    p.add_argument("--root_path", type=str, default=None, help="DomainN/{train,test}/*.h5 tree (default: the reference's)")
    p.add_argument("--snapshot_dir", type=str, default="../model", help="checkpoints go to <snapshot_dir>/<exp>")
    p.add_argument("--synthetic", type=int, default=0, help="N > 0: N synthetic training slices per client instead of HDF5")
    p.add_argument("--dtype", type=str, default=None, help="compute dtype of the HIP kernels (fp32 | bf16 | fp16)")
    p.add_argument("--graph", type=int, default=1, help="capture the training / ALA iteration into a hipGraph")
    p.add_argument("--adamw_frozen", type=str, default="torch2", choices=["torch2", "torch1"],
                   help="AdamW on parameters the FedICRA freeze schedule has switched off: torch2 = skipped (grad None), "
                        "torch1 = the reference's pinned PyTorch 1.10.2 (zeroed gradients: weight decay keeps acting)")
    return p


def check_args(args, world):
    """flower_pCE_2D.py:262-276."""
    from .flower_common import CENTRALIZED_FL, PERSONALIZED_FL
    # flower_common.get_strategy asserts the name (flower_common.py:431-433); a typo must not silently train FedAvg
    assert args.strategy in CENTRALIZED_FL + PERSONALIZED_FL, \
        "unknown --strategy {!r}: one of {}".format(args.strategy, CENTRALIZED_FL + PERSONALIZED_FL)
    assert args.iters > 0
    assert args.eval_iters > 0 and args.eval_iters % args.iters == 0
    assert args.max_iterations > 0 and args.max_iterations % args.eval_iters == 0
    if args.strategy == "FedICRA":
        assert args.iters > args.rep_iters
        assert args.model in ["unet_lc", "unet_lc_multihead"]
    root, ncls, in_chns, sup = DATASETS[args.img_class]
    assert world <= len(sup) or args.synthetic, f"{args.img_class} has {len(sup)} sites"
    args.num_classes, args.in_chns = ncls, in_chns
    args.min_num_clients = world
    if args.root_path is None:
        args.root_path = root
    args.snapshot_path = os.path.join(args.snapshot_dir, args.exp)
    args.patch_size = [args.img_size, args.img_size]
    args.use_graph = bool(args.graph)
    return sup


def round_schedule(args):
    """``range(iters, num_rounds + iters, iters)`` with num_rounds = max_iterations (flower_common.py:256, flower_pCE_2D.py:385)."""
    return range(args.iters, args.max_iterations + args.iters, args.iters)


def make_config(args, server_round, stage):
    return {"iter_global": server_round, "iters": args.iters, "eval_iters": args.eval_iters,
            "batch_size": args.batch_size, "stage": stage}                           # flower_pCE_2D.py:320-338


def state_dict_from_weights(net, weights):
    """``parameters_to_state_dict`` (flower_common.py:236-242) for a DeviceWeights payload: CPU tensors under the
    reference's keys, conv weights in logical [Cout,Cin,k,k] order."""
    from .flat import _as_view
    host, cnt = weights.state.detach().cpu(), weights.counters.cpu()
    sd, ci = OrderedDict(), 0
    for k, v in net.state_dict().items():
        if v.is_floating_point():
            sd[k] = _as_view(host, net._fi_offsets[k], v).contiguous().clone()
        else:
            sd[k] = cnt[ci].clone()
            ci += 1
    return sd


def checkpoint_names(args, iter_num, best_performance=None, client_id=None, client_dice=None):
    """The file names of flower_common.py:343-365 (``best``: the two files of an improved validation Dice; else the
    3000-iteration snapshot)."""
    snap = args.snapshot_path
    if client_id is None:
        if best_performance is None:
            return [os.path.join(snap, "iter_{}.pth".format(iter_num))]
        return [os.path.join(snap, "iter_{}_dice_{}.pth".format(iter_num, round(best_performance, 4))),
                os.path.join(snap, "{}_best_model.pth".format(args.model))]
    if client_dice is None:
        return [os.path.join(snap, "client_{}_iter_{}.pth".format(client_id, iter_num))]
    return [os.path.join(snap, "client_{}_iter_{}_dice_{}.pth".format(client_id, iter_num, round(client_dice, 4))),
            os.path.join(snap, "client_{}_{}_best_model.pth".format(client_id, args.model))]


def synthetic_split(args, cid, n, dense):
    from .synth import phantom_batch
    img, weak, mask = phantom_batch(n, args.img_size, args.in_chns, args.num_classes, cid=cid, index=1 if dense else 0,
                                    labeled_frac=0.1)
    return img, (mask if dense else weak)


def build_client(args, sup_types, rank, device):
    """Model, data and client of rank `rank` (flower_pCE_2D.py:284-318 on the device-resident data path)."""
    from .dataloaders.dataset import BaseDataSets, DeviceLoader, RandomGenerator
    from .flower_common import MyModel
    from .networks import net_factory
    from .networks.unet import set_compute_dtype
    args.cid = rank
    if args.synthetic:
        xi, yi = synthetic_split(args, rank, args.synthetic, dense=False)
        xv, yv = synthetic_split(args, rank, max(4, args.synthetic // 4), dense=True)
        db_train = BaseDataSets.from_arrays(xi, yi, split="train", transform=RandomGenerator(args.patch_size, img_class=args.img_class),
                                            img_class=args.img_class)
        db_val = BaseDataSets.from_arrays(xv, yv, split="val", img_class=args.img_class)
    else:
        client = "client{}".format(rank + 1)
        db_train = BaseDataSets(base_dir=args.root_path, split="train", client=client, sup_type=sup_types[rank],
                                transform=RandomGenerator(args.patch_size, img_class=args.img_class), img_class=args.img_class)
        db_val = BaseDataSets(base_dir=args.root_path, split="val", client=client, img_class=args.img_class)
    trainloader = DeviceLoader(db_train, batch_size=args.batch_size, shuffle=True, device=device)
    valloader = DeviceLoader(db_val, batch_size=1, shuffle=False, device=device)
    net = net_factory(args, net_type=args.model, in_chns=args.in_chns, class_num=args.num_classes)
    if args.dtype:
        set_compute_dtype(net, args.dtype)
    model = MyModel(args, net, trainloader, valloader)
    mod = importlib.import_module("fedicra_amd." + args.procedure)
    client = mod.MyClient(args, model, trainloader, valloader, amp=(args.amp == 1))
    return client, model, net


def _gather(obj, world):
    import torch.distributed as dist
    if world == 1:
        return [obj]
    out = [None] * world
    dist.all_gather_object(out, obj)
    return out


def run(args, log=print):
    """The round loop.  Returns the history of folded validation metrics (rank 0: also writes the checkpoints)."""
    import torch.distributed as dist
    from . import fl
    from .comm import WeightedAllReduce, init_process_group_from_env
    from .flower_common import (PERSONALIZED_FL, VAL_METRICS, DeviceWeights, fit_metrics_aggregation_fn,
                                get_evaluate_metrics_aggregation_fn, get_strategy)
    rank, local, world = init_process_group_from_env()
    sup_types = check_args(args, world)
    device = torch.device("cuda", local)
    torch.cuda.set_device(device)
    random.seed(args.seed)                                       # flower_pCE_2D.py:249-252: every process seeds alike
    np.random.seed(args.seed)
    torch.manual_seed(args.seed)
    if rank == 0:
        os.makedirs(args.snapshot_path, exist_ok=True)
    client, model, net = build_client(args, sup_types, rank, device)
    writer = None
    if rank == 0:
        try:
            from tensorboardX import SummaryWriter
            writer = SummaryWriter(os.path.join(args.snapshot_path, "log"))
        except ImportError:
            pass
    agg = WeightedAllReduce(len(client.trainloader), device=device)
    fold_eval = get_evaluate_metrics_aggregation_fn(args, VAL_METRICS)
    # initial global parameters: one client's (flower_common.py:209); all start from seed 2022, rank 0's are broadcast
    glob = DeviceWeights(net.flat_state.clone(), net.flat_counters.clone())
    if world > 1:
        dist.broadcast(glob.state, src=0)
        dist.broadcast(glob.counters, src=0)
    personalized = args.strategy in PERSONALIZED_FL
    # FedAdagrad / FedAdam / FedYogi: the server optimizer of flwr's FedOpt family runs on the all-reduced mean -- every rank
    # holds the same mean and the same optimizer state, so every rank takes the same step (no server process).  The
    # reference builds these strategies without flwr's mandatory initial_parameters (flower_pCE_2D.py:353-363) and cannot
    # run them as shipped; here they start from the broadcast initial state.
    server_opt = None
    if args.strategy in ("FedAdagrad", "FedAdam", "FedYogi"):
        server_opt = get_strategy(args.strategy, initial_parameters=glob)
    best_performance, history = 0.0, []
    start = timeit.default_timer()
    for current_round in round_schedule(args):
        iter_num = current_round
        res = client.fit(fl.FitIns(parameters=glob, config=make_config(args, current_round, "fit")))
        glob = agg.aggregate(res.parameters)                     # FedAvg.aggregate_fit as one weighted all-reduce
        if server_opt is not None:
            glob = server_opt.server_step(glob, agg.counter_mean)
        else:
            glob = DeviceWeights(glob.state.clone(), glob.counters.clone())   # the reducer reuses its buffers next round
        scalars = {k: v for k, v in res.metrics.items() if not isinstance(v, (bytes, bytearray))}
        fit_metrics = fit_metrics_aggregation_fn(_gather((res.num_examples, scalars), world))
        if writer is not None:
            for cid in range(world):
                for name in TRAIN_SCALAR_METRICS + (["loss_lc"] if args.strategy == "FedICRA" else []):
                    writer.add_scalar("info/client_{}_{}".format(cid, name), fit_metrics["client_{}_{}".format(cid, name)], iter_num)
        if iter_num > 0 and iter_num % args.eval_iters == 0:
            ev = client.evaluate(fl.EvaluateIns(parameters=glob, config=make_config(args, current_round, "evaluate")))
            folded = fold_eval(_gather((ev.num_examples, ev.metrics), world))
            # centralized strategies: the reference scores the global model on the union of the validation sets
            # (evaluate_fn on 'client_all'); the example-weighted fold of the per-client means IS that mean
            mean_metrics = folded
            val_mean_dice = float(mean_metrics["val_mean_dice"])
            history.append((iter_num, {k: float(v) for k, v in folded.items() if k.startswith("val_")}))
            if rank == 0:
                log("iteration {} : ".format(iter_num) + "".join(
                    "mean_{} : {}; ".format(m, mean_metrics["val_mean_{}".format(m)]) for m in VAL_METRICS)
                    + "({:.1f} s)".format(timeit.default_timer() - start))
                if writer is not None:
                    for m in VAL_METRICS:
                        writer.add_scalar("info/val_mean_{}".format(m), mean_metrics["val_mean_{}".format(m)], iter_num)
                        writer.add_scalar("info/val_avg_mean_{}".format(m), folded["val_avg_mean_{}".format(m)], iter_num)
            if val_mean_dice > best_performance:
                best_performance = val_mean_dice
                if rank == 0:
                    state_dict = state_dict_from_weights(net, glob)
                    if not personalized:
                        for path in checkpoint_names(args, iter_num, best_performance):
                            torch.save(state_dict, path)
                    for cid in range(world):                    # per-client files hold the global state (module docstring)
                        dice = float(folded["client_{}_val_mean_dice".format(cid)])
                        for path in checkpoint_names(args, iter_num, client_id=cid, client_dice=dice):
                            torch.save(state_dict, path)
                    log("save model to {}".format(args.snapshot_path))
        if iter_num > 0 and iter_num % 3000 == 0 and rank == 0:
            state_dict = state_dict_from_weights(net, glob)
            if not personalized:
                torch.save(state_dict, checkpoint_names(args, iter_num)[0])
            for cid in range(world):
                torch.save(state_dict, checkpoint_names(args, iter_num, client_id=cid)[0])
        if iter_num >= args.max_iterations:
            break
    if rank == 0:
        log("FL finished in {:.1f} s".format(timeit.default_timer() - start))
    if world > 1:
        dist.barrier()
    return history


def main(argv=None):
    args = build_parser().parse_args(argv)
    run(args)
    import torch.distributed as dist
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
