"""GPU parity of the regulariser losses that follow the hot path against vectors produced by the
reference's own modules (tests/golden) and the CPU oracle on larger seeded inputs."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"

CRF_CASES = {"a": ([{"weight": 1, "xy": 6, "rgb": 0.1}], 5),
             "b": ([{"weight": 0.9, "xy": 6, "rgb": 0.1}, {"weight": 0.1, "xy": 6}], 3)}


def test_gated_crf_matches_reference_golden(golden):
    from fedicra_amd.utils.gate_crf_loss import ModelLossSemsegGatedCRF
    g = golden("g10_gatedcrf.npz")
    for name, (desc, radius) in CRF_CASES.items():
        lg = torch.from_numpy(g[f"{name}/logits"]).to(DEV).requires_grad_(True)
        sample = torch.from_numpy(g[f"{name}/sample"]).to(DEV)
        y = torch.softmax(lg, dim=1)
        y.retain_grad()
        H, W = y.shape[2:]
        loss = ModelLossSemsegGatedCRF()(y, desc, radius, sample, H, W)["loss"]
        loss.backward()
        ref = float(g[f"{name}/loss"])
        assert abs(loss.item() - ref) < 1e-5 * max(1.0, abs(ref)), (name, loss.item(), ref)
        gy = torch.from_numpy(g[f"{name}/grad_y"])
        assert (y.grad.cpu() - gy).abs().max().item() < 1e-5 * gy.abs().max().item() + 1e-9
        gl = torch.from_numpy(g[f"{name}/grad_logits"])
        assert (lg.grad.cpu() - gl).abs().max().item() < 1e-5 * gl.abs().max().item() + 1e-9


def test_gated_crf_trainer_shape_against_oracle():
    """The shape the `_Ours` trainer uses (12 x 2 x 256 x 256, 1-channel image, radius 5) on a quarter-size batch:
    against the loop restatement on the CPU; also ragged sizes (not multiples of the 16-pixel tile)."""
    from fedicra_amd.utils.gate_crf_loss import ModelLossSemsegGatedCRF
    from oracle.gatedcrf_ref import gated_crf_loss
    desc = [{"weight": 1, "xy": 6, "rgb": 0.1}]
    for (N, C, H, W, F_) in [(3, 2, 256, 256, 1), (1, 3, 37, 50, 3)]:
        gen = torch.Generator().manual_seed(H)
        y = torch.softmax(torch.randn(N, C, H, W, generator=gen) * 2, dim=1)
        sample = torch.rand(N, F_, H, W, generator=gen)
        ref, prod = gated_crf_loss(y, desc, 5, sample)
        yd = y.to(DEV).requires_grad_(True)
        loss = ModelLossSemsegGatedCRF()(yd, desc, 5, sample.to(DEV), H, W)["loss"]
        loss.backward()
        assert abs(loss.item() - ref.item()) < 2e-5 * abs(ref.item()), (loss.item(), ref.item())
        gref = -2.0 * prod / (N * H * W)
        assert (yd.grad.cpu() - gref).abs().max().item() < 2e-5 * gref.abs().max().item()
    with pytest.raises(NotImplementedError):
        ModelLossSemsegGatedCRF()(yd, desc, 5, sample.to(DEV), H, W, mask_src=torch.ones(1, 1, H, W, device=DEV))


# ------------------------------------------------------------------------------------------------ tree filter
def _rand_guides(B, C, H, W, seed, ties=False):
    rng = np.random.default_rng(seed)
    if ties:                                           # piecewise-constant: many exactly equal edge weights
        return torch.from_numpy(rng.integers(0, 3, (B, C, H, W)).astype(np.float32))
    return torch.from_numpy(rng.random((B, C, H, W), dtype=np.float32))


@pytest.mark.parametrize("case", [(2, 3, 9, 13, False), (1, 2, 16, 16, True), (3, 1, 20, 7, True), (2, 3, 64, 48, False)])
def test_tree_mst_is_the_references_tree(case):
    """Grid weights bit-equal to torch's, spanning tree = the edge SET the reference's own Boruvka (compiled from its
    source into oracle/_ref) selects -- including inputs full of ties -- and a valid deterministic BFS order."""
    from fedicra_amd.utils.tree_filter import MinimumSpanningTree, TreeFilter2D
    from fedicra_amd import _lib as L
    from oracle import tree_ref as T
    B, C, H, W, ties = case
    fm = _rand_guides(B, C, H, W, seed=H * W, ties=ties)
    wd = torch.empty((B, 2 * H * W - H - W), device=DEV)
    L.tree_grid_weights(fm.to(DEV), wd)
    # same arithmetic (products rounded, then summed in channel order); torch's own reduction order is not specified,
    # so one ulp is allowed here and the spanning trees below are compared on the SAME (device) weights
    assert torch.allclose(wd.cpu(), T.grid_weights(fm), rtol=3e-7, atol=0), "grid weights differ from torch's"
    tree = MinimumSpanningTree(TreeFilter2D.norm2_distance)(fm.to(DEV))
    assert tuple(tree.shape) == (B, H * W - 1, 2)
    idx, wt = T.grid_index(H, W), wd.cpu().numpy()
    edges = tree.edges.cpu().numpy()
    V = H * W
    for b in range(B):
        ref = T.mst_reference(idx, wt[b], V) if T.have_reference_boruvka() else T.mst_kruskal(idx, wt[b], V)
        assert T.edge_set(edges[b]) == T.edge_set(ref), f"image {b}: spanning tree differs from the reference's"
        assert T.edge_set(edges[b]) == T.edge_set(T.mst_kruskal(idx, wt[b], V))
    sidx = torch.empty((B, V), dtype=torch.int32, device=DEV)
    spar, schild = torch.empty_like(sidx), torch.empty((B, V, 4), dtype=torch.int32, device=DEV)
    levels = torch.empty((B, V + 2), dtype=torch.int32, device=DEV)
    import os
    for force_global in ("0", "1"):              # the LDS-resident traversal and the global-memory kernel behind it
        os.environ["FI_TREE_BFS_GLOBAL"] = force_global
        try:
            for t in (sidx, spar, schild, levels):
                t.fill_(-7)
            L.tree_bfs(tree.edges, H, W, sidx, spar, schild, levels)
            torch.cuda.synchronize()
        finally:
            os.environ.pop("FI_TREE_BFS_GLOBAL", None)
        for b in range(B):
            rs, rp, rc, rl = T.bfs(edges[b], V, W)
            assert np.array_equal(sidx[b].cpu().numpy(), rs) and np.array_equal(spar[b].cpu().numpy(), rp)
            assert np.array_equal(schild[b].cpu().numpy(), rc)
            lv = levels[b].cpu().numpy()
            assert lv[0] == len(rl) - 1 and list(lv[1:2 + lv[0]]) == rl


@pytest.mark.parametrize("limits", [None, ("8", "512"), ("4096", "16"), ("4", "8")])
@pytest.mark.parametrize("low_tree", [True, False])
def test_tree_filter_forward_backward_against_oracle(low_tree, limits, monkeypatch):
    if limits is not None:       # shrink the LDS level cache / the streamed chunk: the global-memory fallbacks of the recursions
        monkeypatch.setenv("FI_TREE_CAP", limits[0])
        monkeypatch.setenv("FI_TREE_CHUNK", limits[1])
    _tree_filter_case(low_tree)


def _tree_filter_case(low_tree):
    """TreeFilter2D output, d/d feature and (high-level tree) d/d embedding vs the CPU restatement of refine.cu."""
    from fedicra_amd.utils.tree_filter import MinimumSpanningTree, TreeFilter2D
    from oracle import tree_ref as T
    B, C, Ce, H, W = 2, 2, 3, 24, 20
    g = torch.Generator().manual_seed(5)
    feat = torch.rand(B, C, H, W, generator=g)
    emb = torch.rand(B, Ce, H, W, generator=g) * (0.3 if low_tree else 1.0)
    gout = torch.rand(B, C, H, W, generator=g)
    fr, er = feat.clone().requires_grad_(True), emb.clone().requires_grad_(True)
    fd, ed = feat.to(DEV).requires_grad_(True), emb.to(DEV).requires_grad_(True)
    tree = MinimumSpanningTree(TreeFilter2D.norm2_distance)(ed)
    ref = T.tree_filter(fr, er, tree.edges.cpu().numpy(), 0.02, low_tree)     # same tree: the MST has its own test
    (ref * gout).sum().backward()
    out = TreeFilter2D(groups=1, sigma=0.02)(fd, ed, tree, low_tree=low_tree)
    (out * gout.to(DEV)).sum().backward()
    assert (out.detach().cpu() - ref.detach()).abs().max().item() < 2e-5
    assert (fd.grad.cpu() - fr.grad).abs().max().item() < 2e-5 * max(1.0, fr.grad.abs().max().item())
    if low_tree:
        assert ed.grad is None and er.grad is None
    else:
        assert (ed.grad.cpu() - er.grad).abs().max().item() < 5e-5 * max(1.0, er.grad.abs().max().item())


def test_tree_energy_losses_against_oracle():
    """TreeEnergyLoss / MScaleRecurveTreeEnergyLoss (flower_common.py:646-818): loss value, filtered maps and the
    gradients that reach the logits and the three guidance feature maps, vs the CPU restatement run on the SAME trees
    (the restatement recomputes them from the same device-side weights' order: piecewise-smooth guides without near-ties)."""
    from fedicra_amd.tree_energy import MScaleRecurveTreeEnergyLoss, TreeEnergyLoss
    from oracle import tree_ref as T
    B, C, H, W = 2, 2, 32, 24
    g = torch.Generator().manual_seed(11)
    preds = torch.randn(B, C, H, W, generator=g)
    img = torch.rand(B, 3, H, W, generator=g)
    highs = [torch.rand(B, 2, H // s, W // s, generator=g) for s in (4, 2, 1)]       # aux heads at S/4, S/2, S
    rois = (torch.rand(B, H, W, generator=g) < 0.9)
    pr = preds.clone().requires_grad_(True)
    hr = [h.clone().requires_grad_(True) for h in highs]
    ref = T.mscale_recurve_tree_energy_loss(pr, img, hr[0], hr[1], hr[2], rois, 0.4)
    ref[0].backward()
    pd = preds.to(DEV).requires_grad_(True)
    hd = [h.to(DEV).requires_grad_(True) for h in highs]
    out = MScaleRecurveTreeEnergyLoss()(pd, img.to(DEV), hd[0], hd[1], hd[2], rois.to(DEV), 0.4)
    out[0].backward()
    assert abs(out[0].item() - ref[0].item()) < 2e-5 * max(1.0, abs(ref[0].item())), (out[0].item(), ref[0].item())
    for k in range(1, 4):
        assert (out[k].detach().cpu() - ref[k].detach()).abs().max().item() < 5e-5
    assert (pd.grad.cpu() - pr.grad).abs().max().item() < 5e-5 * max(1.0, pr.grad.abs().max().item())
    for a, b_ in zip(hd, hr):
        assert (a.grad.cpu() - b_.grad).abs().max().item() < 2e-4 * max(1e-3, b_.grad.abs().max().item())
    # single-tree variant, low-level tree only
    p2 = preds.to(DEV).requires_grad_(True)
    l2, AS = TreeEnergyLoss()(p2, img.to(DEV), None, rois.to(DEV), 1.0)
    r2, ASr = T.tree_energy_loss(preds.clone().requires_grad_(True), img, rois, 1.0)
    assert abs(l2.item() - r2.item()) < 2e-5 and (AS.detach().cpu() - ASr.detach()).abs().max().item() < 5e-5


def test_ours_procedure_first_iteration_against_oracle_and_graph_replay():
    """flower_pCE_2D_GateCRFMsacleTreeEnergyLoss_Ours.MyClient: the loss terms of the first iteration (pCE, tree energy,
    gated CRF) vs the CPU oracle evaluated on the oracle network with the same seeded state and dropout masks; then the
    captured hipGraph must reproduce the eager run."""
    import argparse
    from fedicra_amd import ops
    from fedicra_amd.flower_common import MyModel
    from fedicra_amd.flower_pCE_2D_GateCRFMsacleTreeEnergyLoss_Ours import MyClient
    from fedicra_amd.flower_pCE_2D import _GraphStep
    from fedicra_amd.networks.unet import UNet_MultiHead, set_compute_dtype
    from oracle import tree_ref as T
    from oracle.gatedcrf_ref import gated_crf_loss
    from oracle.losses_ref import pce_loss
    from oracle.unet_ref import RefUNet, seeded_state
    from helpers import loader
    batches = loader(2, 4, 64, cid=0)

    def args_(**kw):
        a = argparse.Namespace(strategy="FedAvg", amp=0, model="unet_multihead", cid=0, min_num_clients=1, num_classes=2,
                               img_class="faz", base_lr=0.01, max_iterations=30000, iters=4, rep_iters=3, alpha=0.5,
                               snapshot_path=None, use_graph=False, tree_loss_weight=0.1)
        a.__dict__.update(kw)
        return a

    def mk():
        m = UNet_MultiHead(1, 2)
        seeded_state(m, 2022)
        return set_compute_dtype(m.cuda(), "fp32")

    # ---- oracle: forward in train mode with pinned masks, the three loss terms
    b = batches[0]
    ref = RefUNet(1, 2, heads=3)
    seeded_state(ref, 2022)
    ref.train()
    torch.manual_seed(3)
    x = b["image"].unsqueeze(1)
    o = ref(x)
    ce_ref = pce_loss(o[0], b["label"], 2)
    tree_ref = T.mscale_recurve_tree_energy_loss(o[0], x.repeat(1, 3, 1, 1), o[6], o[7], o[8], b["label"] == 2, 0.1)[0]
    crf_ref = gated_crf_loss(torch.softmax(o[0], 1), [{"weight": 1, "xy": 6, "rgb": 0.1}], 5, x)[0]
    # ---- HIP, eager, same masks
    net = mk()
    args = args_()
    client = MyClient(args, MyModel(args, net, batches, batches), batches, batches)
    client.model.train()
    client._ensure_optimizer().reset_round()
    ops.set_dropout_mask_provider(lambda shape, p: torch.empty(shape).bernoulli_(1 - p))
    try:
        torch.manual_seed(3)
        rec = _GraphStep()
        client._iteration(*client._stage(b), rec)
    finally:
        ops.set_dropout_mask_provider(None)
    assert abs(rec.loss_ce.item() - ce_ref.item()) < 1e-5
    assert abs(rec.loss_crf.item() - crf_ref.item()) < 2e-5 * max(1.0, abs(crf_ref.item()))
    assert abs(rec.loss_tree.item() - tree_ref.item()) < 1e-4 * max(1.0, abs(tree_ref.item())), (rec.loss_tree.item(), tree_ref.item())
    # ---- graph replay == eager over 4 iterations (device RNG dropout)
    finals, losses = [], []
    for use_graph in (False, True):
        a2 = args_(use_graph=use_graph)
        ops.manual_seed(1)
        net2 = mk()
        c2 = MyClient(a2, MyModel(a2, net2, batches, batches), batches, batches)
        c2._train({"iter_global": 4, "iters": 4, "eval_iters": 8, "batch_size": 4, "stage": "fit"})
        finals.append(net2.flat_state.clone())
        losses.append(list(c2.last_losses))
    assert all(np.isfinite(losses[0])) and np.allclose(losses[0][:2], losses[1][:2], atol=2e-5), (losses[0], losses[1])
    assert np.allclose(losses[0], losses[1], atol=5e-3)


def test_readme_training_configuration_runs_captured_and_eager():
    """The reference README's command (`--procedure ..._Ours --model unet_lc_multihead --strategy FedICRA --alpha 1
    --rep_iters 3`): pCE + multi-scale tree energy + 0.1 gated CRF + LC loss (no-grad forwards with the other clients'
    embeddings), head-only then body-only freeze phases -- one captured hipGraph per phase equals eager launches."""
    import argparse
    from fedicra_amd import ops
    from fedicra_amd.flower_common import MyModel
    from fedicra_amd.flower_pCE_2D_GateCRFMsacleTreeEnergyLoss_Ours import MyClient
    from fedicra_amd.networks import net_factory
    from fedicra_amd.networks.unet import set_compute_dtype
    from helpers import loader
    K, cid = 3, 1
    batches = loader(2, 4, 64, cid=cid, device=DEV)
    finals, losses = [], []
    for use_graph in (False, True):
        args = argparse.Namespace(strategy="FedICRA", amp=0, model="unet_lc_multihead", cid=cid, min_num_clients=K,
                                  num_classes=2, img_class="faz", base_lr=0.01, max_iterations=30000, iters=6, rep_iters=3,
                                  alpha=1.0, snapshot_path=None, use_graph=use_graph, tree_loss_weight=0.1)
        torch.manual_seed(2022)
        ops.manual_seed(4)
        net = net_factory(args, net_type="unet_lc_multihead", in_chns=1, class_num=2).cuda()
        set_compute_dtype(net, "fp32")
        head0 = net.decoder.out_conv.weight.detach().clone()
        enc0 = net.encoder.in_conv.conv_conv[0].weight.detach().clone()
        c = MyClient(args, MyModel(args, net, batches, batches), batches, batches)
        c._train({"iter_global": 60, "iters": 6, "eval_iters": 99, "batch_size": 4, "stage": "fit"})
        c._train({"iter_global": 61, "iters": 6, "eval_iters": 99, "batch_size": 4, "stage": "fit"})
        assert not torch.equal(head0, net.decoder.out_conv.weight) and not torch.equal(enc0, net.encoder.in_conv.conv_conv[0].weight)
        finals.append(net.flat_state.clone())
        losses.append(list(c.last_losses))
        if use_graph:
            assert set(c._steps) == {"head", "body"} and all(r.graph is not None for r in c._steps.values())
    assert np.isfinite(losses[0]).all() and np.isfinite(losses[1]).all()
    assert np.allclose(losses[0][:2], losses[1][:2], atol=5e-5), (losses[0], losses[1])
    assert np.allclose(losses[0], losses[1], atol=2e-2), (losses[0], losses[1])


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_lc_probe_forward_has_the_side_effects_of_the_full_forward(dtype):
    """`model(x, j, heatmap_only=True)` (the LC loss's no-grad forwards) against the full forward from the same state and
    dropout seeds: identical heat-map and trunk outputs, the same model state afterwards (to one ulp) -- every BatchNorm
    running statistic and counter, the heads' included -- while the heads' tensors and the logits are not produced."""
    import argparse
    from fedicra_amd import ops
    from fedicra_amd.networks import net_factory
    from fedicra_amd.networks.unet import set_compute_dtype
    from helpers import loader
    x = loader(1, 4, 64, cid=1, device=DEV)[0]["image"].unsqueeze(1)
    args = argparse.Namespace(min_num_clients=4, cid=1)
    outs, states = [], []
    for probe in (False, True):
        torch.manual_seed(2022)
        ops.manual_seed(3)
        net = net_factory(args, net_type="unet_lc_multihead", in_chns=1, class_num=2).cuda().train()
        set_compute_dtype(net, dtype)
        off = torch.zeros(1, dtype=torch.int32, device=DEV)
        ops.set_dropout_seed_offset(off)
        try:
            with torch.no_grad():
                ops.begin_iteration(x.device)
                o = net(x, 2, heatmap_only=probe)
        finally:
            ops.set_dropout_seed_offset(None)
        outs.append(o)
        states.append((net.flat_state.clone(), net.flat_counters.clone()))
    full, probe = outs
    assert probe[0] is None and probe[7] is None and probe[8] is None and probe[9] is None and full[9] is not None
    assert torch.equal(full[6][-1], probe[6][-1])
    for i in range(2, 6):
        assert torch.equal(full[i], probe[i])
    assert torch.equal(states[0][1], states[1][1]) and int(states[1][1].min()) == 1
    # the statistics are fp64 atomic sums: their order differs between a storing and a non-storing launch, which can move
    # a running statistic by an fp32 ulp or two (more where var << mean^2 cancels)
    assert torch.allclose(states[0][0], states[1][0], rtol=2e-6, atol=1e-7), float((states[0][0] - states[1][0]).abs().max())
    with torch.no_grad():                                # eval mode / autograd on: the flag is ignored
        net.eval()
        assert net(x, 2, heatmap_only=True)[0] is not None
