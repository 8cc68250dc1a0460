"""Round-3 parity tests: the evaluation path and the FedAvg counter path on the device against the reference's own
functions (golden g8 / g6), AdamW's torch-1 frozen-parameter semantics, FusedSGD on a 3D model."""
import argparse

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


def test_g8_device_evaluate_matches_the_references_evaluate(golden):
    """fedicra_amd.flower_common.evaluate (device arg-max + integer counts + device hd95, chunked forward) on the logits
    and label maps of golden g8 = the reference's own flower_common.evaluate / val_2D.test_single_volume with the restated
    medpy formulas: every key, every value."""
    from fedicra_amd.flower_common import evaluate
    g = golden("g8_eval_metrics.npz")
    for name, ncls in (("faz", 2), ("odoc", 3)):
        table = torch.from_numpy(g[f"{name}/logits"]).to(DEV)

        class Net(torch.nn.Module):
            def __init__(self):
                super().__init__()
                self.dummy = torch.nn.Parameter(torch.zeros(1, device=DEV))

            def forward(self, x):
                idx = x.reshape(x.shape[0], -1)[:, 0].round().long()
                return [table[idx].contiguous(memory_format=torch.channels_last)]

        images = g[f"{name}/images"]
        batches = [{"image": torch.from_numpy(images[i:i + 1]), "label": torch.from_numpy(g[f"{name}/gt"][i:i + 1])}
                   for i in range(images.shape[0])]

        class Loader(list):
            pass

        loader = Loader(batches)
        loader.dataset = list(range(len(batches)))
        args = argparse.Namespace(num_classes=ncls, img_class=name)
        out = evaluate(args, Net(), loader)
        assert list(out.keys()) == [str(k) for k in g[f"{name}/keys"]]
        np.testing.assert_allclose(np.array([float(v) for v in out.values()]), g[f"{name}/vals"], rtol=0, atol=1e-14)


def test_g6_device_aggregate_reproduces_the_references_counter_path(golden):
    """aggregate_device + WeightedAllReduce (one process, K co-located clients) on the K client states of golden g6: the
    counters after the int64 sum -> float64 divide -> truncation are the ones the reference's set_weights loaded; the fp32
    state is the numpy aggregate bit for bit."""
    from fedicra_amd.comm import WeightedAllReduce
    from fedicra_amd.flower_common import DeviceWeights, aggregate_device
    from fedicra_amd.networks.unet import UNet
    from oracle.unet_ref import seeded_state
    g = golden("g6_fedavg_counters.npz")
    n_all = [int(v) for v in g["n_all"]]
    for K in (2, 5, 8):
        nets = []
        for k in range(K):
            net = UNet(1, 2)
            seeded_state(net, 100 + k)
            net = net.to(DEV)
            net.flat_counters.copy_(torch.tensor([7 * k + 3 * j + 1 for j in range(net.flat_counters.numel())]))
            nets.append(net)
        dws = [DeviceWeights(n.flat_state, n.flat_counters) for n in nets]
        got = aggregate_device([(dw, n_all[k]) for k, dw in enumerate(dws)])
        np.testing.assert_array_equal(got.counters.cpu().numpy(), g[f"K{K}/counters_loaded"])
        recv = UNet(1, 2).to(DEV)
        recv.flat_state.copy_(got.state)
        np.testing.assert_array_equal(recv.state_dict()["decoder.out_conv.weight"].cpu().numpy(), g[f"K{K}/out_conv_weight"])
        red = WeightedAllReduce(n_all[:K], device=torch.device(DEV))
        got2 = red.aggregate(dws)
        torch.cuda.synchronize()
        np.testing.assert_array_equal(got2.counters.cpu().numpy(), g[f"K{K}/counters_loaded"])
        assert torch.equal(got2.state, got.state)


@pytest.mark.parametrize("frozen", ["torch1", "torch2"])
def test_fused_adamw_frozen_parameter_semantics_against_torch_adamw(frozen):
    """FedICRA's freeze schedule under the two zero_grad semantics (LOG.md section 5, 'a stated deviation'): FusedAdamW against
    torch.optim.AdamW on the CPU fed the same gradients for two rounds of 3 head-phase + 2 body-phase iterations, a fresh
    optimizer per round (flower_pCE_2D.py:55).  torch1 = PyTorch 1.10.2 of the reference's environment: a parameter that has
    held a gradient keeps a zero gradient while frozen, so AdamW keeps decaying it; torch2 = gradients set to None: skipped."""
    from fedicra_amd.networks.unet import UNet
    from fedicra_amd.optim import FusedAdamW
    from oracle.unet_ref import seeded_state
    torch.manual_seed(5)
    net = UNet(1, 2)
    seeded_state(net, 11)
    ref = {n: p.detach().clone().requires_grad_(True) for n, p in net.named_parameters()}
    net = net.to(DEV)
    names = [n for n, _ in net.named_parameters()]
    head = {"decoder.out_conv.weight", "decoder.out_conv.bias"}
    opt = FusedAdamW(net, lr=0.01, base_lr=0.01, max_iterations=1000, frozen=frozen)
    params = dict(net.named_parameters())
    gen = torch.Generator().manual_seed(9)
    for rnd in range(2):
        opt.reset_round()
        opt.set_lr(0.01, 0)
        topt = torch.optim.AdamW(list(ref.values()), lr=0.01, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2)
        for it in range(5):
            active = head if it < 3 else set(names) - head
            opt.zero_grad()
            for n in names:
                if n in active:
                    g = torch.randn(ref[n].shape, generator=gen) * 1e-2
                    ref[n].grad = g.clone()
                    params[n].grad = g.to(DEV)
                elif frozen == "torch1" and ref[n].grad is not None:
                    ref[n].grad = torch.zeros_like(ref[n])          # zero_grad() of torch 1.x: zeroed in place, never None
                else:
                    ref[n].grad = None
            opt.step()
            topt.step()
            torch.cuda.synchronize()
            worst = max(float((params[n].detach().cpu() - ref[n].detach()).abs().max()) for n in names)
            assert worst < 1e-6, (frozen, rnd, it, worst)                     # fp32 round-off of 10 steps; the two semantics differ by ~1e-4
    if frozen == "torch1":
        # the two semantics really differ on this schedule: the body decayed during the head phase of round 2
        assert len(opt._ever) == len(names)


def test_fused_sgd_steps_a_3d_model_whose_gradients_arrive_outside_the_flat_sink():
    """unet_3D hands autograd ordinary gradient tensors; FusedSGD must move them into the flat gradient buffer
    before the fused step (the reference's 3D trainers use SGD).  One step against torch.optim.SGD on the CPU."""
    from fedicra_amd.networks.net_factory_3d import net_factory_3d
    from fedicra_amd.optim import FusedSGD
    torch.manual_seed(3)
    net = net_factory_3d("unet_3D", in_chns=1, class_num=2)     # (the reference's factory moves the model to the GPU itself)
    ref = {n: p.detach().cpu().clone().requires_grad_(True) for n, p in net.named_parameters()}
    net = net.to(DEV)
    opt = FusedSGD(net, lr=0.01, base_lr=0.01, max_iterations=100)
    topt = torch.optim.SGD(list(ref.values()), lr=0.01, momentum=0.9, weight_decay=1e-4)
    gen = torch.Generator().manual_seed(4)
    for _ in range(2):
        opt.zero_grad()
        for n, p in net.named_parameters():
            g = torch.randn(ref[n].shape, generator=gen) * 1e-2
            ref[n].grad = g.clone()
            p.grad = g.to(DEV)                                        # an ordinary tensor, not the parameter's sink view
        opt.step()
        topt.step()
    torch.cuda.synchronize()
    for n, p in net.named_parameters():
        assert float((p.detach().cpu() - ref[n].detach()).abs().max()) < 1e-6, n


@pytest.mark.parametrize("name", ["FedAdagrad", "FedAdam", "FedYogi"])
def test_round_driver_applies_the_fedopt_server_step_to_the_all_reduced_mean(name, tmp_path):
    """``--strategy FedAdam`` must not silently train FedAvg.  (a) ``FedOpt.server_step`` on the mean coming
    out of the weighted all-reduce == ``aggregate_fit`` of the same strategy on the clients' results, two rounds;
    (b) the round driver accepts the strategy and runs it; an unknown name is rejected before any work."""
    import os
    import subprocess
    import sys
    from fedicra_amd import fl
    from fedicra_amd.comm import WeightedAllReduce
    from fedicra_amd.flower_common import DeviceWeights, get_strategy
    g = torch.Generator().manual_seed(21)
    n = 50001
    init = DeviceWeights(torch.randn(n, generator=g).to(DEV), torch.tensor([3, 9], dtype=torch.int64, device=DEV))
    a = get_strategy(name, initial_parameters=DeviceWeights(init.state.clone(), init.counters.clone()))
    b = get_strategy(name, initial_parameters=DeviceWeights(init.state.clone(), init.counters.clone()))
    n_k = [21, 13, 17]
    red = WeightedAllReduce(n_k, device=torch.device(DEV))
    for rnd in range(2):
        ws = [DeviceWeights((init.state + 0.1 * torch.randn(n, generator=g).to(DEV)), init.counters + 5 * (k + rnd + 1))
              for k in range(3)]
        results = [(None, fl.FitRes(status=fl.Status("OK", "Success"), parameters=w, num_examples=n_k[k], metrics={}))
                   for k, w in enumerate(ws)]
        want, _ = a.aggregate_fit(rnd + 1, results, [])
        mean = red.aggregate(ws)
        got = b.server_step(mean, red.counter_mean)
        torch.cuda.synchronize()
        assert torch.equal(got.state, want.state) and torch.equal(got.counters, want.counters), (name, rnd)
        assert not torch.equal(got.state, mean.state)               # it is not the plain mean
    if name != "FedAdam":
        return
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    base = [sys.executable, "-m", "fedicra_amd.run_federated", "--exp", "t_opt", "--model", "unet", "--img_class", "faz",
            "--iters", "2", "--eval_iters", "4", "--max_iterations", "4", "--batch_size", "4", "--img_size", "64",
            "--synthetic", "8", "--snapshot_dir", str(tmp_path), "--graph", "0"]
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    out = subprocess.run(base + ["--strategy", "FedAdam"], env=env, cwd=root, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "FL finished" in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]
    bad = subprocess.run(base + ["--strategy", "Fedicra"], env=env, cwd=root, capture_output=True, text=True, timeout=600)
    assert bad.returncode != 0 and "unknown --strategy" in bad.stderr


def test_ala_epoch_without_the_auxiliary_heads_and_with_half_dgrads_gives_the_same_weights():
    """FedICRA's ALA loop (flower_common.py:566-602) on the device: (a) the copy's auxiliary head is skipped (nothing of it is
    read, its BatchNorm statistics are discarded with the copy) and (b) the first convolution of every UpBlock computes only
    the gradient half that is asked for (the skip half comes from the frozen encoder).  Both must leave the mixing weights
    and the decoder exactly as the full computation does."""
    from fedicra_amd import ops
    from fedicra_amd.flower_common import DeviceWeights, MyModel
    from fedicra_amd.networks import net_factory
    from helpers import loader
    outs = []
    for skip_aux in (True, False):
        args = argparse.Namespace(strategy="FedICRA", amp=0, model="unet_lc", cid=1, min_num_clients=3, num_classes=2,
                                  img_class="faz", base_lr=0.01, max_iterations=200, iters=4, rep_iters=1, alpha=1.0,
                                  snapshot_path=None, use_graph=False)
        torch.manual_seed(2022)
        ops.manual_seed(7)
        net = net_factory(args, net_type="unet_lc", in_chns=1, class_num=2).to(DEV)
        batches = loader(2, 4, 64, cid=1, device=DEV)
        model = MyModel(args, net, batches, batches)
        model.train()
        model.start_phase = False
        model.verbose = False
        model.ala_skip_aux = skip_aux
        model.ala_batched = False                                  # (the per-batch loop on both sides: same dropout draws)
        g = torch.Generator().manual_seed(5)
        glob = DeviceWeights(net.flat_state + 0.02 * torch.randn(net.flat_state.shape, generator=g).to(DEV), net.flat_counters.clone())
        model.set_weights(glob, {"iter_global": 60})
        torch.cuda.synchronize()
        outs.append((model.fedaa_weights.clone(), net.flat_params.clone(), list(model.ala_epoch_losses)))
    assert outs[0][2] == outs[1][2]
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    assert float((outs[0][0] < 1).float().mean()) > 0.01          # the epoch really moved the mixing weights


def test_dgrad_of_one_half_of_a_concatenated_input_equals_that_half_of_the_full_dgrad():
    from fedicra_amd import ops
    import torch.nn as nn
    torch.manual_seed(3)
    for dt in (torch.float32, torch.bfloat16):
        conv = nn.Conv2d(48, 32, 3, padding=1).to(DEV)
        bn = nn.BatchNorm2d(32).to(DEV)
        a = torch.randn(2, 20, 24, 16, device=DEV).to(dt)
        b = torch.randn(2, 20, 24, 32, device=DEV).to(dt)
        grads = {}
        for need in ("both", "x0", "x1"):
            a_ = a.clone().requires_grad_(need in ("both", "x0"))
            b_ = b.clone().requires_grad_(need in ("both", "x1"))
            conv.weight.requires_grad_(False), conv.bias.requires_grad_(False)
            ops.begin_iteration(a.device)
            z = ops.conv_bn_act(a_, b_, conv, bn, 0.01)
            z.float().square().sum().backward()
            grads[need] = (a_.grad, b_.grad)
        assert torch.equal(grads["x0"][0], grads["both"][0]) and grads["x0"][1] is None
        assert torch.equal(grads["x1"][1], grads["both"][1]) and grads["x1"][0] is None


def test_pcs_gate_and_lc_total_kernels_against_the_module_arithmetic():
    """fi_pcs_gate_fwd / _bwd against the reference module's expression sigmoid(fc2(cat(avg, e)) + fc2(cat(max, e))),
    e = fc1(onehot) (unet.py:122-141) in torch on the CPU -- values and the gradients w.r.t. avg / max; fi_lc_loss_fwd / _bwd
    against -mean_g mse(h, o_g) and its place in the total (flower_pCE_2D.py:128-139)."""
    import torch.nn as nn
    from fedicra_amd import ops
    torch.manual_seed(9)
    B, Cc, K = 5, 256, 8
    fc1 = nn.Sequential(nn.Conv2d(K, Cc, 1, bias=False), nn.ReLU(), nn.Conv2d(Cc, Cc, 1, bias=False))
    fc2 = nn.Sequential(nn.Conv2d(2 * Cc, Cc // 16, 1, bias=False), nn.ReLU(), nn.Conv2d(Cc // 16, Cc, 1, bias=False))
    avg = torch.randn(B, Cc, requires_grad=True)
    mx = torch.randn(B, Cc, requires_grad=True)
    who = torch.tensor([3, 0, 7, 3, 1])
    emb = torch.zeros(B, K)
    emb[torch.arange(B), who] = 1
    e = fc1(emb.reshape(B, K, 1, 1))
    ref = torch.sigmoid(fc2(torch.cat([avg.reshape(B, Cc, 1, 1), e], 1)) + fc2(torch.cat([mx.reshape(B, Cc, 1, 1), e], 1))).reshape(B, Cc)
    gout = torch.randn(B, Cc)
    ref.backward(gout)
    a2, m2 = avg.detach().to(DEV).requires_grad_(True), mx.detach().to(DEV).requires_grad_(True)
    got = ops.pcs_gate(a2, m2, who.to(torch.int32).to(DEV), fc1.to(DEV), fc2.to(DEV))
    got.backward(gout.to(DEV))
    assert float((got.detach().cpu() - ref.detach()).abs().max()) < 2e-6
    assert float((a2.grad.cpu() - avg.grad).abs().max()) < 2e-6 and float((m2.grad.cpu() - mx.grad).abs().max()) < 2e-6
    # LC total
    G, n = 7, 12 * 256
    h = torch.rand(12, 256, 1, 1, requires_grad=True)
    others = torch.rand(G, n)
    ce = torch.tensor(0.7, requires_grad=True)
    acc = 0
    for g_ in range(G):
        acc = acc + torch.nn.functional.mse_loss(h, others[g_].reshape(h.shape))
    lc_ref = -acc / G
    tot_ref = torch.add(ce, lc_ref, alpha=0.5)
    (tot_ref * 3.0).backward()
    h2, ce2 = h.detach().to(DEV).requires_grad_(True), ce.detach().to(DEV).requires_grad_(True)
    tot, lc = ops.lc_total(ce2, h2, others.to(DEV).reshape(-1), G, 0.5)
    (tot * 3.0).backward()
    assert abs(float(tot.detach()) - float(tot_ref.detach())) < 1e-6 and abs(float(lc) - float(lc_ref.detach())) < 1e-6 and not lc.requires_grad
    assert float((h2.grad.cpu() - h.grad).abs().max()) < 1e-9 and abs(float(ce2.grad) - 3.0) < 1e-7


@pytest.mark.parametrize("use_graph", [False, True])
def test_lc_forwards_beside_the_own_forward_leave_the_state_of_the_in_line_order(use_graph):
    """The K-1 no-grad LC forwards on a second stream beside the client's own forward (flower_pCE_2D._iteration) against the
    in-line order of the reference (own forward, then the loop, flower_pCE_2D.py:106,128-139): same losses, same parameters,
    same BatchNorm running statistics and counters after a round of head- and body-phase iterations -- the per-layer events
    keep the order-dependent running-statistics recursion in the reference's order (fp64 statistic sums are atomic: fp32
    round-off apart)."""
    from fedicra_amd import ops
    from fedicra_amd.flower_common import MyModel
    from fedicra_amd.flower_pCE_2D import MyClient
    from fedicra_amd.networks import net_factory
    from helpers import loader
    res = []
    for beside in (False, True):
        args = argparse.Namespace(strategy="FedICRA", amp=0, model="unet_lc", cid=1, min_num_clients=4, num_classes=2,
                                  img_class="faz", base_lr=0.01, max_iterations=200, iters=5, rep_iters=2, alpha=1.0,
                                  snapshot_path=None, use_graph=use_graph)
        torch.manual_seed(2022)
        ops.manual_seed(11)
        net = net_factory(args, net_type="unet_lc", in_chns=1, class_num=2).to(DEV)
        batches = loader(3, 4, 64, cid=1, device=DEV)
        model = MyModel(args, net, batches, batches)
        client = MyClient(args, model, batches, batches)
        client.probe_beside = beside
        cfg = {"iter_global": 60, "iters": 5, "eval_iters": 10, "batch_size": 4, "stage": "fit"}
        client._train(cfg)
        client._train(cfg)                                   # second round: the captured steps replay
        torch.cuda.synchronize()
        res.append((list(client.last_losses), net.flat_state.clone(), net.flat_counters.clone()))
    (l0, s0, c0), (l1, s1, c1) = res
    assert torch.equal(c0, c1)
    assert np.allclose(l0, l1, rtol=0, atol=2e-5), (l0, l1)
    assert torch.allclose(s0, s1, rtol=1e-4, atol=2e-5), float((s0 - s1).abs().max())


def test_batch_stager_hands_out_the_batches_it_was_given_pinned_ahead_and_pageable_serially():
    """staging.BatchStager: pinned host batches are copied ahead on the side stream (several in flight, ring re-use), pageable ones
    serially at fetch(); either way the consumer sees the batch it asked for."""
    from fedicra_amd.staging import BatchStager
    g = torch.Generator().manual_seed(11)
    st = BatchStager(DEV, slots=4)
    mk = lambda pin: {"image": (lambda t: t.pin_memory() if pin else t)(torch.randn(3, 32, 32, generator=g)),
                      "label": (lambda t: t.pin_memory() if pin else t)(torch.randint(0, 3, (3, 32, 32), generator=g).to(torch.uint8))}
    pinned = [mk(True) for _ in range(9)]
    for i, b in enumerate(pinned):
        for nxt in pinned[i + 1:i + 3]:
            st.prefetch(nxt)
        x, y = st.fetch(b)
        xc, yc = x.clone(), y.clone()
        st.release()
        torch.cuda.synchronize()
        assert torch.equal(xc.cpu(), b["image"]) and torch.equal(yc.cpu(), b["label"]), i
    assert st.h2d_bytes == sum(b["image"].numel() * 4 + b["label"].numel() for b in pinned)
    pageable = [mk(False) for _ in range(3)]
    for i, b in enumerate(pageable):
        st.prefetch(pageable[(i + 1) % 3])
        assert not st._pending                                       # nothing goes ahead from pageable memory
        x, y = st.fetch(b)
        xc, yc = x.clone(), y.clone()
        st.release()
        torch.cuda.synchronize()
        assert torch.equal(xc.cpu(), b["image"]) and torch.equal(yc.cpu(), b["label"]), i
