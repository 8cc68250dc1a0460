"""GPU parity of the HBM-resident data path: the augmentation gather against the CPU oracle
(numpy rot90/flip + the scipy-pinned nearest-neighbour rotation), the reference-generated fixture, and a torch
DataLoader(num_workers=0) over the oracle transform.  Bit-exact throughout (byte / index work)."""
import random

import numpy as np
import pytest
import torch

from oracle import augment_ref as A

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _set(cls_, n, shape, seed):
    rng = np.random.default_rng(seed)
    imgs = rng.random((n,) + shape, dtype=np.float32)
    labs = rng.integers(0, A.ROT_CVAL[cls_][1] + 1, (n,) + shape[-2:]).astype(np.uint8)
    return imgs, labs


@pytest.mark.parametrize("cls_,shape", [("faz", (256, 256)), ("faz", (33, 33)), ("odoc", (3, 48, 48)), ("polyp", (3, 20, 20))])
def test_augment_gather_every_parameter_combination(cls_, shape):
    """All 4 quarter turns x 2 flip axes x {no rotation, every angle in [-45, 45)} -- one launch, vs the oracle."""
    from fedicra_amd.dataloaders import BaseDataSets, RandomGenerator
    imgs, labs = _set(cls_, 5, shape, 3)
    gen = RandomGenerator(list(shape[-2:]), cls_)
    res = BaseDataSets.from_arrays(imgs, labs, img_class=cls_).resident(DEV)
    draws = [(k, ax, ang) for k in (-1, 0, 1, 2, 3) for ax in (0, 1) for ang in [None] + list(range(-45, 45))
             if not (k == -1 and ax == 1)]
    srcs = [i % 5 for i in range(len(draws))]
    out_i, out_l = gen.apply(res, srcs, draws)
    out_i, out_l = out_i.cpu().numpy(), out_l.cpu().numpy()
    for b, (s, d) in enumerate(zip(srcs, draws)):
        wi, wl = A.apply(imgs[s], labs[s], cls_, *d)
        np.testing.assert_array_equal(out_i[b], wi, err_msg=str(d))
        np.testing.assert_array_equal(out_l[b], wl, err_msg=str(d))


def test_augment_non_square_rotation_and_half_turn():
    from fedicra_amd.dataloaders import BaseDataSets, RandomGenerator
    imgs, labs = _set("faz", 2, (40, 28), 4)
    gen = RandomGenerator([40, 28], "faz")
    res = BaseDataSets.from_arrays(imgs, labs, img_class="faz").resident(DEV)
    draws = [(k, ax, ang) for k in (-1, 0, 2) for ax in (0, 1) for ang in (None, -45, -13, 0, 7, 44)]
    out_i, out_l = gen.apply(res, [b % 2 for b in range(len(draws))], draws)
    for b, d in enumerate(draws):
        wi, wl = A.apply(imgs[b % 2], labs[b % 2], "faz", *d)
        np.testing.assert_array_equal(out_i[b].cpu().numpy(), wi)
        np.testing.assert_array_equal(out_l[b].cpu().numpy(), wl)
    with pytest.raises(ValueError):
        gen.apply(res, [0], [(3, 0, None)])


def test_random_generator_reproduces_the_reference_fixture(golden):
    """Per-sample protocol, the reference's seeds -> the reference's RandomGenerator outputs (g11)."""
    from fedicra_amd.dataloaders import RandomGenerator
    g = golden("g11_augment.npz")
    for name in ("faz", "odoc", "polyp"):
        random.seed(int(g[f"{name}/seeds"][0]))
        np.random.seed(int(g[f"{name}/seeds"][1]))
        gen = RandomGenerator(list(g[f"{name}/image"].shape[-2:]), name)
        for i in range(g[f"{name}/image"].shape[0]):
            out = gen({"image": g[f"{name}/image"][i], "label": g[f"{name}/label"][i]})
            assert out["image"].dtype == torch.float32 and out["label"].dtype == torch.uint8
            np.testing.assert_array_equal(out["image"].cpu().numpy(), g[f"{name}/out_image"][i])
            np.testing.assert_array_equal(out["label"].cpu().numpy(), g[f"{name}/out_label"][i])


def test_device_loader_equals_a_torch_dataloader_over_the_oracle_transform():
    """Shuffled, augmented epochs: DeviceLoader vs torch DataLoader(shuffle=True, num_workers=0) whose dataset applies
    the oracle transform -- same torch / python / numpy seeds, two epochs, ragged last batch."""
    from torch.utils.data import DataLoader, Dataset
    from fedicra_amd.dataloaders import BaseDataSets, DeviceLoader, RandomGenerator
    imgs, labs = _set("faz", 29, (64, 64), 9)

    class RefSet(Dataset):
        def __len__(self):
            return len(imgs)

        def __getitem__(self, i):
            out, _ = A.random_generator({"image": imgs[i], "label": labs[i]}, "faz")
            return {"image": torch.from_numpy(out["image"]), "label": torch.from_numpy(out["label"]), "idx": i}

    def seed():
        torch.manual_seed(3)
        random.seed(4)
        np.random.seed(5)

    seed()
    want = [b for _ in range(2) for b in DataLoader(RefSet(), batch_size=12, shuffle=True, num_workers=0)]
    seed()
    ds = BaseDataSets.from_arrays(imgs, labs, split="train", transform=RandomGenerator([64, 64], "faz"), img_class="faz")
    loader = DeviceLoader(ds, batch_size=12, shuffle=True, device=DEV)
    got = [b for _ in range(2) for b in loader]
    assert len(loader) == 3 and len(got) == len(want) == 6 and got[2]["image"].shape == (5, 64, 64)
    for a, b in zip(got, want):
        assert a["idx"].tolist() == b["idx"].tolist()
        assert torch.equal(a["image"].cpu(), b["image"]) and torch.equal(a["label"].cpu(), b["label"])
    val = DeviceLoader(BaseDataSets.from_arrays(imgs, labs, split="val", img_class="faz"), batch_size=1, device=DEV)
    first = next(iter(val))
    assert len(val) == 29 and torch.equal(first["image"].cpu(), torch.from_numpy(imgs[:1]))


def test_trainer_runs_on_a_device_loader():
    """flower_pCE_2D.MyClient._train fed by the resident loader (stands where the DataLoader stands, :303-306): the
    captured training step consumes the loader's device batches; the loss falls on a learnable phantom."""
    import argparse
    from fedicra_amd import ops
    from fedicra_amd.dataloaders import BaseDataSets, DeviceLoader, RandomGenerator
    from fedicra_amd.flower_common import MyModel
    from fedicra_amd.flower_pCE_2D import MyClient
    from fedicra_amd.networks.unet import UNet
    from fedicra_amd.synth import phantom_batch
    x, y, _ = phantom_batch(24, 64, 1, 2, cid=0, labeled_frac=0.2)
    ds = BaseDataSets.from_arrays(x, y, split="train", transform=RandomGenerator([64, 64], "faz"), img_class="faz")
    args = argparse.Namespace(strategy="FedAvg", amp=0, model="unet", cid=0, min_num_clients=1, num_classes=2,
                              img_class="faz", base_lr=0.01, max_iterations=30000, iters=12, rep_iters=3, alpha=0.5,
                              snapshot_path=None, use_graph=True)
    ops.manual_seed(1)
    torch.manual_seed(0)
    net = UNet(1, 2).cuda()
    tl = DeviceLoader(ds, batch_size=12, shuffle=True, device=DEV)
    client = MyClient(args, MyModel(args, net, tl, tl), tl, tl)
    loss0, _ = client._train({"iter_global": 0, "iters": 2, "eval_iters": 1, "batch_size": 12, "stage": 1})
    loss1, _ = client._train({"iter_global": 1, "iters": 12, "eval_iters": 1, "batch_size": 12, "stage": 1})
    assert np.isfinite(loss0) and np.isfinite(loss1) and loss1 < loss0


def test_colocated_clients_train_exactly_like_clients_alone():
    """Two clients sharing the GPU (one HIP stream + device context + captured step each, iterations interleaved by
    flower_pCE_2D.train_colocated) end bit-identical to the same clients trained one after the other: nothing a
    captured step bakes in (accumulator arena, dropout counter) is shared between them."""
    import argparse
    from fedicra_amd import ops
    from fedicra_amd.flower_common import MyModel
    from fedicra_amd.flower_pCE_2D import MyClient, train_colocated
    from fedicra_amd.networks.unet import UNet, set_compute_dtype
    from oracle.unet_ref import seeded_state
    from helpers import loader

    def make(cid, dtype):
        args = argparse.Namespace(strategy="FedAvg", amp=0, model="unet", cid=cid, min_num_clients=2, num_classes=2,
                                  img_class="faz", base_lr=0.01, max_iterations=30000, iters=6, rep_iters=3, alpha=0.5,
                                  snapshot_path=None, use_graph=True)
        net = UNet(1, 2)
        seeded_state(net, 2022 + cid)
        net = net.cuda()
        set_compute_dtype(net, dtype)
        batches = loader(3, 4, 64, cid=cid, device=DEV)
        return MyClient(args, MyModel(args, net, batches, batches), batches, batches)

    cfg = lambda r: {"iter_global": r, "iters": 6, "eval_iters": 60, "batch_size": 4, "stage": "fit"}
    for dtype in ("fp32", "bf16"):
        ops.manual_seed(5)
        alone = [make(0, dtype), make(1, dtype)]
        res_alone = [[c._train(cfg(r)) for r in range(2)] for c in alone]      # eager, capture, then replays
        ops.manual_seed(5)
        together = [make(0, dtype), make(1, dtype)]
        res_tog = [train_colocated(together, [cfg(r), cfg(r)]) for r in range(2)]
        torch.cuda.synchronize()
        assert together[0].stream is not None and together[0].stream != together[1].stream
        assert together[0].ctx.arena.buf.data_ptr() != together[1].ctx.arena.buf.data_ptr()
        for k in range(2):
            assert torch.equal(alone[k]._net().flat_state, together[k]._net().flat_state), (dtype, k)
            assert alone[k].last_losses == together[k].last_losses
            assert res_alone[k][1][0] == res_tog[1][k][0]
        assert not torch.equal(together[0]._net().flat_state, together[1]._net().flat_state)


def test_ala_captured_iteration_matches_reference_golden_and_the_eager_path(golden):
    """FedICRA set_weights (a4) with the ALA batch captured into a hipGraph (args.use_graph): (1) eval mode, against the
    reference's own set_weights (golden g7) on the first call (converge loop), a second call (cached model + graph reused
    across rounds) and the identical-global early-out; (2) train mode with the device dropout RNG: bit-identical to the
    eager path, ALA epoch losses included."""
    from fedicra_amd import ops
    from fedicra_amd.flower_common import MyModel
    from fedicra_amd.networks.unet import UNet_LC
    from oracle.unet_ref import RefUNetLC, seeded_state
    from oracle import fed_ref
    from helpers import assert_ck, loader
    from test_unet_gpu import _args, _mk
    g = golden("g7_ala.npz")
    K, cid = 3, 1
    batches = loader(3, 4, 64, cid=cid)
    donor = RefUNetLC(1, 2, 1, K, K, cid)
    seeded_state(donor, 200)
    glob = fed_ref.get_weights(donor)
    seeded_state(donor, 300)
    glob2 = fed_ref.get_weights(donor)
    args = _args(strategy="FedICRA", model="unet_lc", cid=cid, min_num_clients=K, use_graph=True)
    net = _mk(UNet_LC, 1, 2, 1, K, K, cid, seed=100, lc=True).eval()
    model = MyModel(args, net, batches, batches)
    model.eval()
    model.set_weights(glob, {"iter_global": 60})
    assert model._ala["graph"] is not None and len(model.ala_epoch_losses) == int(g["eval/first_epochs"])
    assert len(net.state_dict()) == 144, "the cached ALA copy must not leak into the wire format"
    for k, v in net.state_dict().items():
        assert_ck(v.double().cpu(), g["eval/first/" + k], rtol=2e-4, atol=2e-5, what=k)
    graph = model._ala["graph"]
    model.set_weights(glob2, {"iter_global": 70})
    assert model._ala["graph"] is graph and len(model.ala_epoch_losses) == 1
    for k, v in net.state_dict().items():
        assert_ck(v.double().cpu(), g["eval/second/" + k], rtol=2e-4, atol=2e-5, what=k)
    model.set_weights(glob2, {"iter_global": 80})
    assert_ck(net.state_dict()["decoder.out_conv.weight"].cpu(), g["eval/third_out_conv_ck"], rtol=1e-6, atol=1e-7)
    finals = []
    for use_graph in (False, True):
        ops.manual_seed(9)
        a = _args(strategy="FedICRA", model="unet_lc", cid=cid, min_num_clients=K, use_graph=use_graph)
        n = _mk(UNet_LC, 1, 2, 1, K, K, cid, seed=100, lc=True).train()
        m = MyModel(a, n, batches, batches)
        m.train()
        m.start_phase = False                                  # one epoch per call
        m.set_weights(glob, {"iter_global": 60})
        m.set_weights(glob2, {"iter_global": 70})
        finals.append((n.flat_state.clone(), list(m.ala_epoch_losses), m.fedaa_weights.clone()))
    assert torch.equal(finals[0][0], finals[1][0]) and finals[0][1] == finals[1][1]
    assert torch.equal(finals[0][2], finals[1][2]) and 0.0 <= float(finals[1][2].min()) and float(finals[1][2].max()) <= 1.0


def test_wire_format_roundtrip_on_the_device_uses_the_flat_store():
    """a2/a3: get_weights = the reference's list of arrays in state_dict order (values, shapes, dtypes), produced from
    ONE device-to-host copy of the flat state; set_weights(list) pairs arrays with keys like load_state_dict(strict=
    False), truncates float64 counters to int64, rejects a wrong shape, and uploads with one copy."""
    import argparse
    import time
    from fedicra_amd.flower_common import MyModel
    from fedicra_amd.networks.unet import UNet_LC_MultiHead
    torch.manual_seed(1)
    net = UNet_LC_MultiHead(1, 2, 1, 8, 8, 3).cuda()
    args = argparse.Namespace(strategy="FedAvg", amp=0, cid=3, num_classes=2, img_class="faz")
    mm = MyModel(args, net, [], [])
    sd = {k: v.detach().cpu().numpy() for k, v in net.state_dict().items()}
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ws = mm.get_weights(None)
    t1 = time.perf_counter()
    assert len(ws) == len(sd) == 160
    for a, (k, b) in zip(ws, sd.items()):
        assert a.shape == b.shape and a.dtype == b.dtype and np.array_equal(a, b), k
    new = [np.asarray(a * 0.5 + 1.0, dtype=np.float64) if a.dtype == np.float32 else np.float64(5.9) for a in ws]
    t2 = time.perf_counter()
    mm.set_weights(new, {})
    torch.cuda.synchronize()
    t3 = time.perf_counter()
    for (k, v), a in zip(net.state_dict().items(), new):
        want = torch.from_numpy(a).to(v.dtype) if v.is_floating_point() else torch.tensor(5, dtype=torch.int64)
        assert torch.equal(v.cpu(), want), k
    print(f"get_weights {1e3 * (t1 - t0):.2f} ms, set_weights {1e3 * (t3 - t2):.2f} ms (160 arrays, 9.4 MB)")
    bad = list(new)
    bad[0] = bad[0][:-1]
    with pytest.raises(RuntimeError, match="size mismatch"):
        mm.set_weights(bad, {})
    mm.set_weights(new[:10], {})                               # strict=False: a short list loads its prefix only


def test_bench_multi_rank_path_on_one_gpu_via_gloo():
    """bench.py launched the way the driver launches it for N > 1 (torch.distributed.run, one rank per GPU), except that
    both ranks sit on cuda:0 and exchange through gloo (a 1-GPU box cannot host two RCCL ranks): rank-dependent client
    ids, FedICRA rounds (training under the freeze schedule, weighted all-reduce on a side stream with the absent
    clients' constant term, set_weights with its ALA epoch), barriers, max-over-ranks timing and the single JSON line
    from rank 0 are all exercised (configs[2] at 128^2 to keep two time-sliced ranks quick)."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, FEDICRA_DIST_BACKEND="gloo", FEDICRA_FORCE_DEVICE="0")
    port = str(29600 + os.getpid() % 300)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", port, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "20", "--warmup", "10",
           "--no-roofline", "--no-fp32", "--size", "128", "--loader-batches", "2"]
    out = subprocess.run(cmd, env=env, cwd=root, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 20 and d["config"]["clients_hosted"] == 2 and d["scaling"] == "weak"
    assert d["value"] > 0 and abs(d["value"] - 20 * 12 * 2 / (d["ms_per_step"] * 20 / 1e3)) < 0.01 * d["value"]
    assert "cpu_baseline" not in d                     # rank 0 at N = 1 only


def test_mini_federation_dice_against_the_cpu_oracle():
    """BASELINE metric (3), end to end: 2 FedAvg clients x 3 rounds x 8 local iterations (fresh AdamW per round, poly LR,
    weighted aggregation with n_k = #batches, global state loaded into both clients), then `evaluate` -- the HIP path in
    fp32 against the CPU oracle driven through the same rounds with the same dropout masks.  Round 1 is held to fp32
    parity; after 48 AdamW steps the reference's own round-off sensitivity (DESIGN.md "parity bar") bounds what two
    correct implementations can share: the final Dice values are printed and only required to show that both learned."""
    import argparse
    from fedicra_amd import ops
    from fedicra_amd.flower_common import MyModel, aggregate_device, evaluate
    from fedicra_amd.flower_pCE_2D import MyClient
    from fedicra_amd.networks.unet import UNet, set_compute_dtype
    from fedicra_amd.synth import phantom_batch
    from oracle import fed_ref
    from oracle.losses_ref import eval_case
    from oracle.unet_ref import RefUNet, seeded_state
    K, rounds, iters, n_k = 2, 3, 8, [3, 2]
    data = []
    for cid in range(K):
        bs = []
        for i in range(n_k[cid]):
            img, weak, _ = phantom_batch(4, 64, 1, 2, cid=cid, index=i, labeled_frac=0.3)
            bs.append({"image": torch.from_numpy(img), "label": torch.from_numpy(weak)})
        data.append(bs)
    vimg, _, vmask = phantom_batch(8, 64, 1, 2, cid=7, dense=True)
    val = [{"image": torch.from_numpy(vimg[i:i + 1]), "label": torch.from_numpy(vmask[i:i + 1])} for i in range(8)]

    # ---- CPU oracle
    refs = [RefUNet(1, 2) for _ in range(K)]
    for r in refs:
        seeded_state(r, 2022)
    states = [fed_ref.TrainState(0.01) for _ in range(K)]
    ref_losses = []
    for rnd in range(rounds):
        res = []
        for cid in range(K):
            torch.manual_seed(100 * rnd + cid)
            _, met = fed_ref.local_train(refs[cid], states[cid], data[cid], iters=iters, num_classes=2, base_lr=0.01,
                                         max_iterations=200)
            res.append((fed_ref.get_weights(refs[cid]), n_k[cid]))
            ref_losses.append(float(met["loss"][-1]) if hasattr(met["loss"], "__len__") else float(met["loss"]))
        glob = fed_ref.fedavg_aggregate(res)
        for r in refs:
            fed_ref.set_weights_plain(r, glob)
    refs[0].eval()
    ref_dice = 0.0
    with torch.no_grad():
        for b in val:
            pred = refs[0](b["image"].unsqueeze(1))[0].argmax(1)[0].numpy()
            ref_dice += eval_case(pred, b["label"][0].numpy(), 2)[0]
    ref_dice /= len(val)

    # ---- HIP path
    clients = []
    for cid in range(K):
        args = argparse.Namespace(strategy="FedAvg", amp=0, model="unet", cid=cid, min_num_clients=K, num_classes=2,
                                  img_class="faz", base_lr=0.01, max_iterations=200, iters=iters, rep_iters=3, alpha=0.5,
                                  snapshot_path=None, use_graph=False)
        net = UNet(1, 2)
        seeded_state(net, 2022)
        net = net.cuda()
        set_compute_dtype(net, "fp32")
        clients.append(MyClient(args, MyModel(args, net, data[cid], data[cid]), data[cid], data[cid]))
    ops.set_dropout_mask_provider(lambda shape, p: torch.empty(shape).bernoulli_(1 - p))
    hip_losses = []
    try:
        for rnd in range(rounds):
            res = []
            for cid in range(K):
                torch.manual_seed(100 * rnd + cid)
                last, _ = clients[cid]._train({"iter_global": rnd, "iters": iters, "eval_iters": 99, "batch_size": 4,
                                               "stage": "fit"})
                hip_losses.append(last)
                res.append((clients[cid].model.get_device_weights(), n_k[cid]))
            glob = aggregate_device(res)
            for c in clients:
                c.model.set_weights(glob, {"iter_global": rnd})
    finally:
        ops.set_dropout_mask_provider(None)
    met = evaluate(clients[0].args, clients[0].model.model, val)
    assert torch.equal(clients[0]._net().flat_state, clients[1]._net().flat_state)      # both hold the global state
    for a, b in zip(fed_ref.get_weights(refs[0]), clients[0].model.get_weights(None)):
        assert a.shape == b.shape and a.dtype == b.dtype
    nbt = int(clients[0]._net().flat_counters[0])
    assert nbt == int(refs[0].state_dict()["encoder.in_conv.conv_conv.1.num_batches_tracked"]), nbt
    print(f"mini federation: val_mean_dice HIP {met['val_mean_dice']:.4f} vs CPU oracle {ref_dice:.4f}; last losses per "
          f"(round, client) HIP {np.round(hip_losses, 4).tolist()} oracle {np.round(ref_losses, 4).tolist()}")
    assert abs(hip_losses[0] - ref_losses[0]) < 5e-3 and abs(hip_losses[1] - ref_losses[1]) < 5e-3       # round 1
    # Round 1 is held to fp32 parity above.  What the END point can be held to is no longer judged here on one 3-round run
    # (the reference's own Dice after 3 rounds moves between 0.17 and 0.99 with the dropout seed, the thread count or a 1e-6
    # perturbation of its initial weights): tests/test_round4_gpu.py holds (i) the measured horizon of north_star's 1e-4 bound
    # against the reference's own vectors and (ii) the converged 12-round Dice of five seeds against the reference's own
    # 16-run spread (golden g19).  Here: both sides trained and produced a valid metric.
    assert 0.0 <= ref_dice <= 1.0 and 0.0 <= met["val_mean_dice"] <= 1.0


def test_many_captured_clients_in_one_process():
    """A simulation that builds fresh clients every round captures a new training step each time: the pinned tables the
    captured wgrad reductions read from are grown outside the captures (ops.reserve_graph_tables), not a fixed pool."""
    import argparse
    from fedicra_amd import ops
    from fedicra_amd.flower_common import MyModel
    from fedicra_amd.flower_pCE_2D import MyClient
    from fedicra_amd.networks.unet import UNet
    from helpers import loader
    batches = loader(1, 2, 32, cid=0, device=DEV)
    last = None
    for k in range(36):                                        # the old fixed pool had 32 slots
        args = argparse.Namespace(strategy="FedAvg", amp=0, model="unet", cid=0, min_num_clients=1, num_classes=2,
                                  img_class="faz", base_lr=0.01, max_iterations=30000, iters=3, rep_iters=3, alpha=0.5,
                                  snapshot_path=None, use_graph=True)
        net = UNet(1, 2).cuda()
        c = MyClient(args, MyModel(args, net, batches, batches), batches, batches)
        loss, _ = c._train({"iter_global": 0, "iters": 3, "eval_iters": 9, "batch_size": 2, "stage": "fit"})
        assert c._steps["all"].graph is not None and np.isfinite(loss)
        last = c
    assert last is not None


@pytest.mark.parametrize("name", ["FedAdagrad", "FedAdam", "FedYogi"])
def test_fedopt_device_path_equals_host_path(name):
    """FedOpt on DeviceWeights (fi_fedopt_step over the flat state) against the same strategy on numpy payloads: three
    rounds, fp32 state bit for bit (numpy's evaluation order and roundings), counters after the clients' truncation."""
    from fedicra_amd import fl
    from fedicra_amd.flower_common import DeviceWeights, get_strategy
    n = 100003
    g = torch.Generator().manual_seed(3)
    init = torch.randn(n, generator=g)
    cnt0 = torch.tensor([5, 9], dtype=torch.int64)

    class Res:
        def __init__(self, params, ne):
            self.parameters, self.num_examples, self.metrics = params, ne, {}

    host = get_strategy(name, initial_parameters=fl.ndarrays_to_parameters([init.numpy().copy(), np.array(5), np.array(9)]))
    dev = get_strategy(name, initial_parameters=DeviceWeights(init.to(DEV), cnt0.to(DEV)))
    for rnd in range(3):
        ws = [torch.randn(n, generator=g) * (0.5 + rnd) for _ in range(2)]
        cs = [torch.tensor([7 + rnd, 11], dtype=torch.int64), torch.tensor([8 + 2 * rnd, 3], dtype=torch.int64)]
        hres = [(None, Res(fl.ndarrays_to_parameters([w.numpy(), np.array(int(c[0])), np.array(int(c[1]))]), ne))
                for w, c, ne in zip(ws, cs, (3, 5))]
        dres = [(None, Res(DeviceWeights(w.to(DEV), c.to(DEV)), ne)) for w, c, ne in zip(ws, cs, (3, 5))]
        hp, _ = host.aggregate_fit(rnd, hres, [])
        dp, _ = dev.aggregate_fit(rnd, dres, [])
        hw = fl.parameters_to_ndarrays(hp)
        assert torch.equal(dp.state.cpu(), torch.from_numpy(hw[0])), float((dp.state.cpu() - torch.from_numpy(hw[0])).abs().max())
        assert dp.counters.tolist() == [int(np.float64(hw[1])), int(np.float64(hw[2]))]


def test_device_fedavg_equals_numpy_aggregate_bit_for_bit():
    """aggregate_device (fi_scale / fi_axpy over flat device states) against flwr's `aggregate` restated in numpy, on
    random data, K = 2, 5, 8 with the FAZ-like n_k: bit-identical fp32 (products rounded before the sum -- an FMA in the
    accumulate differs in ~25 % of the elements), counters after the float64 divide + truncation."""
    from fedicra_amd.flower_common import DeviceWeights, aggregate, aggregate_device
    g = torch.Generator().manual_seed(11)
    n = 200001
    n_k = [21, 13, 17, 59, 3, 21, 13, 17]
    for K in (2, 5, 8):
        ws = [torch.randn(n, generator=g) * (1 + 0.3 * k) for k in range(K)]
        cs = [torch.tensor([10 * k + 3, 7 + k], dtype=torch.int64) for k in range(K)]
        want = aggregate([([w.numpy(), np.array(int(c[0])), np.array(int(c[1]))], n_k[k]) for k, (w, c) in enumerate(zip(ws, cs))])
        got = aggregate_device([(DeviceWeights(w.to(DEV), c.to(DEV)), n_k[k]) for k, (w, c) in enumerate(zip(ws, cs))])
        assert torch.equal(got.state.cpu(), torch.from_numpy(want[0])), K
        assert got.counters.tolist() == [int(want[1]), int(want[2])]


def test_base_datasets_getitem_with_transform_matches_reference_golden(golden, tmp_path, monkeypatch):
    """BaseDataSets.__getitem__ with the RandomGenerator transform over the dict-backed release tree (g16: the reference's
    own classes on the same tree under the same python / numpy seeds)."""
    import os
    import sys
    import types
    from fedicra_amd.dataloaders import BaseDataSets, RandomGenerator
    from oracle.dataset_tree import dataset_tree
    g = golden("g16_base_datasets.npz")
    store = {}
    dataset_tree(str(tmp_path), store)

    class File(dict):
        def __init__(self, path, mode="r"):
            super().__init__(store[os.path.normpath(path)])

        def __enter__(self):
            return self

        def __exit__(self, *a):
            return False

    monkeypatch.setitem(sys.modules, "h5py", types.SimpleNamespace(File=File))
    ds = BaseDataSets(str(tmp_path), "train", RandomGenerator([16, 16], "faz"), "client2", "scribble", "faz")
    order = np.argsort(ds.sample_list)
    assert [ds.sample_list[int(i)] for i in order] == [str(n) for n in g["aug/names"]]
    random.seed(16)
    np.random.seed(16)
    outs = [ds[int(i)] for i in order]
    np.testing.assert_array_equal(np.stack([o["image"].cpu().numpy() for o in outs]), g["aug/images"])
    np.testing.assert_array_equal(np.stack([o["label"].cpu().numpy() for o in outs]), g["aug/labels"])


def test_run_federated_round_driver_two_ranks_on_one_gpu(tmp_path):
    """fedicra_amd.run_federated launched like the driver launches bench.py (torch.distributed.run, one rank per client; both
    ranks on cuda:0 exchanging through gloo): rounds of fit -> weighted all-reduce -> set_weights (ALA) -> evaluate with the
    launcher's flags, the all-gathered metric folds, and rank 0's checkpoints under the reference's names holding the
    reference's state_dict keys (flower_common.py:343-365, consumed by code/test.py:263-265)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, FEDICRA_DIST_BACKEND="gloo", FEDICRA_FORCE_DEVICE="0")
    port = str(29900 + os.getpid() % 90)
    for strategy, model, extra in (("FedAvg", "unet", []), ("FedICRA", "unet_lc", ["--alpha", "1.0", "--rep_iters", "1"])):
        exp = "t_" + strategy
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
               "127.0.0.1", "--master-port", port, "-m", "fedicra_amd.run_federated", "--exp", exp, "--strategy", strategy,
               "--model", model, "--img_class", "faz", "--iters", "2", "--eval_iters", "4", "--max_iterations", "8",
               "--batch_size", "4", "--img_size", "64", "--synthetic", "12", "--snapshot_dir", str(tmp_path), "--graph", "0"] + extra
        out = subprocess.run(cmd, env=env, cwd=root, capture_output=True, text=True, timeout=900)
        assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
        assert "iteration 4 : mean_dice" in out.stdout and "iteration 8 : mean_dice" in out.stdout and "FL finished" in out.stdout
        names = sorted(os.listdir(os.path.join(tmp_path, exp)))
        assert f"client_0_{model}_best_model.pth" in names and f"client_1_{model}_best_model.pth" in names, names
        assert any(n.startswith("client_1_iter_") and "_dice_" in n for n in names)
        if strategy == "FedAvg":
            assert "unet_best_model.pth" in names and any(n.startswith("iter_") and "_dice_" in n for n in names)
        else:
            assert not any(n.startswith("iter_") for n in names)           # personalised: no central model is saved
        sd = torch.load(os.path.join(tmp_path, exp, f"client_0_{model}_best_model.pth"))
        from fedicra_amd.networks import net_factory
        import argparse
        ref_keys = list(net_factory(argparse.Namespace(min_num_clients=2, cid=0), net_type=model, in_chns=1, class_num=2).state_dict())
        assert list(sd) == ref_keys and sd["encoder.in_conv.conv_conv.0.weight"].shape == (16, 1, 3, 3)
        assert sd["encoder.in_conv.conv_conv.1.num_batches_tracked"].dtype == torch.int64
