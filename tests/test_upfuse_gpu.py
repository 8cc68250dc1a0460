"""fi_conv1x1_up2x_fwd (csrc/upfuse.hip): UpBlock's conv1x1 + bilinear x2 (/root/reference/code/networks/unet.py:57-59,65-67) as
one launch, against the two launches it replaces, against an fp64 evaluation of the same operands, and through autograd."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _two_launch(x, coef, slope, groups, wp, bias, cout):
    """The path the fused kernel replaces, on the library's own kernels."""
    from fedicra_amd import _lib as L
    N, h, w, cin = x.shape
    y = torch.empty((N, h, w, cout), dtype=x.dtype, device=x.device)
    if coef is None:
        L.conv2d_fwd(x, None, wp, bias, y, None, None, ksize=1)
    else:
        L.conv2d_fwd_fused(x, L.in_xform(coef, slope), None, None, wp, bias, y, None, ksize=1, groups=groups, cout=cout)
    u = torch.empty((N, 2 * h, 2 * w, cout), dtype=x.dtype, device=x.device)
    L.upsample2x_fwd(y, u)
    return y, u


def _ref64(x, coef, slope, groups, w, bias, dtype):
    """fp64: z = act(BN(x)) rounded to the storage type, conv1x1 + bias rounded to the storage type, bilinear align_corners."""
    N = x.shape[0]
    z = x.double()
    if coef is not None:
        gi = N // groups
        sc = coef[0].double().repeat_interleave(gi, 0)[:, None, None, :]
        sh = coef[1].double().repeat_interleave(gi, 0)[:, None, None, :]
        t = (x.float() * sc.float() + sh.float())                 # fp32 like the kernels (one fma vs mul+add: < 1 ulp of fp32)
        z = torch.maximum(t, t * slope).to(dtype).double()
    y = torch.einsum("nhwc,oc->nhwo", z, w.double()) + bias.double()
    y = y.to(dtype).double()
    u = F.interpolate(y.permute(0, 3, 1, 2), scale_factor=2, mode="bilinear", align_corners=True)
    return u.permute(0, 2, 3, 1)


CASES = [  # cin, cout, N, h, w, groups (0 = plain source)
    (32, 16, 2, 16, 32, 0), (32, 16, 6, 13, 24, 3), (64, 32, 4, 8, 16, 2), (128, 64, 2, 8, 8, 0), (128, 64, 4, 6, 16, 2),
    (256, 128, 2, 4, 4, 0), (256, 128, 2, 8, 8, 0), (32, 32, 2, 5, 12, 0), (64, 64, 3, 1, 8, 3), (32, 16, 1, 2, 2, 0),
    (64, 32, 2, 32, 64, 2), (32, 16, 2, 64, 128, 2),
]


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("case", CASES)
def test_fused_equals_two_launches_and_fp64(case, dtype):
    from fedicra_amd import _lib as L
    cin, cout, N, h, w, groups = case
    g = torch.Generator(device="cpu").manual_seed(cin * 1000 + h * 10 + w)
    x = torch.randn(N, h, w, cin, generator=g).to(DEV).to(dtype)
    wt = (torch.randn(cout, cin, generator=g) / cin ** 0.5).to(DEV)
    bias = torch.randn(cout, generator=g).to(DEV)
    wp = wt.to(dtype).contiguous()
    coef, slope = None, 0.01
    if groups:
        coef = torch.stack([1.0 + 0.3 * torch.randn(groups, cin, generator=g), 0.2 * torch.randn(groups, cin, generator=g)]).to(DEV)
    y, u2 = _two_launch(x, coef, slope, max(groups, 1), wp, bias, cout)
    ref = _ref64(x, coef, slope, max(groups, 1), wp, bias, dtype)
    eps = 2.0 ** -8 if dtype == torch.bfloat16 else 2.0 ** -11
    tol = 3 * eps * max(1.0, float(ref.abs().max()))
    assert float((u2.double() - ref).abs().max()) <= tol          # the baseline itself
    for rows in (0, 1, 2, 3):
        L.upfuse_tuning(rows)
        try:
            u = torch.full((N, 2 * h, 2 * w, cout), float("nan"), dtype=dtype, device=DEV)
            t0 = None if coef is None else L.in_xform(coef, slope)
            assert L.conv1x1_up2x_fwd(x, t0, wp, bias, u, groups=max(groups, 1))
            torch.cuda.synchronize()
        finally:
            L.upfuse_tuning(0)
        assert not torch.isnan(u.float()).any(), f"rows={rows}: output not fully written"
        assert float((u.double() - ref).abs().max()) <= tol, f"rows={rows}"
        # against the two launches: the interpolation is identical operand for operand, the convolution may accumulate its
        # channel chunks in another order -> at most the last bit of a stored element, on few elements
        diff = (u.float() - u2.float()).abs()
        nbad = int((diff > 0).sum())
        assert float(diff.max()) <= 2 * eps * max(1.0, float(ref.abs().max())), f"rows={rows}"
        assert nbad <= 0.02 * u.numel(), f"rows={rows}: {nbad} of {u.numel()} elements differ from the two-launch form"


def test_unsupported_shapes_are_declined():
    from fedicra_amd import _lib as L
    x = torch.randn(1, 4, 4, 48, device=DEV).bfloat16()
    u = torch.empty(1, 8, 8, 16, device=DEV, dtype=torch.bfloat16)
    assert L.conv1x1_up2x_fwd(x, None, torch.zeros(16, 48, device=DEV, dtype=torch.bfloat16), None, u) is False
    xf = torch.randn(1, 4, 4, 32, device=DEV)
    uf = torch.empty(1, 8, 8, 16, device=DEV)
    assert L.conv1x1_up2x_fwd(xf, None, torch.zeros(16, 32, device=DEV), None, uf) is False


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_autograd_matches_the_two_ops(dtype):
    """ops.conv1x1_up against ops.upsample2x(ops.conv2d(...)): same backward kernels, so the gradients agree bit for bit
    wherever the forward does not enter (dx, dw, db depend on du and x only)."""
    from fedicra_amd import ops
    torch.manual_seed(3)
    conv = torch.nn.Conv2d(64, 32, 1).to(DEV)
    x = torch.randn(2, 16, 16, 64, device=DEV).to(dtype)
    du = torch.randn(2, 32, 32, 32, device=DEV).to(dtype)
    outs = []
    for fused in (True, False):
        conv.weight.grad = conv.bias.grad = None
        xi = x.clone().requires_grad_(True)
        ops.begin_iteration(torch.device(DEV))
        u = ops.conv1x1_up(xi, conv) if fused else ops.upsample2x(ops.conv2d(xi, None, conv))
        u.backward(du)
        ops.flush_wgrad()
        torch.cuda.synchronize()
        outs.append((u.detach().float(), xi.grad.float(), conv.weight.grad.clone(), conv.bias.grad.clone()))
    a, b = outs
    eps = 2.0 ** -8 if dtype == torch.bfloat16 else 1e-6
    assert float((a[0] - b[0]).abs().max()) <= 2 * eps * max(1.0, float(b[0].abs().max()))
    for i in (1, 2, 3):
        assert torch.equal(a[i], b[i]), i


def test_probe_path_moves_the_same_statistics():
    """The batched no-grad LC forward with the fused up-sampling against FI_UPFUSE=0 (two launches): the heat-maps and the state
    (BatchNorm running statistics) the forwards leave behind agree to the storage type's round-off."""
    import argparse
    from fedicra_amd import ops
    from fedicra_amd.networks import net_factory
    from fedicra_amd.networks.unet import set_compute_dtype
    from helpers import loader
    x = loader(1, 4, 64, cid=1, in_chns=3, ncls=3, device=DEV)[0]["image"]
    K, cid = 5, 1
    args = argparse.Namespace(min_num_clients=K, cid=cid)
    others = [j for j in range(K) if j != cid]
    res = []
    for fused in (True, False):
        ops._UPFUSE = fused
        try:
            torch.manual_seed(2022)
            ops.manual_seed(3)
            net = net_factory(args, net_type="unet_lc", in_chns=3, class_num=3).cuda().train()
            set_compute_dtype(net, "bf16")
            ctx = ops.new_context()
            ctx.seed_offset = torch.full((1,), 7, dtype=torch.int32, device=DEV)
            with ops.use_context(ctx), torch.no_grad():
                ops.begin_iteration(x.device)
                net(x)
                hm = net.probe_heatmaps(x, others)
                assert hm is not None
            torch.cuda.synchronize()
            res.append(([h.float().clone() for h in hm], net.flat_state.clone()))
        finally:
            ops._UPFUSE = True
    (h0, s0), (h1, s1) = res
    for a, b in zip(h0, h1):
        assert torch.equal(a, b)                       # the heat-maps come out of the encoder: untouched by the decoder's form
    assert torch.allclose(s0, s1, rtol=5e-3, atol=1e-3), float((s0 - s1).abs().max())


@pytest.mark.parametrize("use_graph", [False, True])
def test_probe_decoder_beside_the_backward_pass_leaves_the_same_state(use_graph):
    """The LC forwards' decoder half only moves BatchNorm statistics; flower_pCE_2D._iteration lets it run on beside the LC loss
    and the backward pass and joins it before the optimizer step.  Against the join before the LC loss: the same losses, the
    same parameters / running statistics / counters after two rounds of head- and body-phase iterations, eager and captured."""
    import argparse
    import numpy as np
    from fedicra_amd import ops
    from fedicra_amd.flower_common import MyModel
    from fedicra_amd.flower_pCE_2D import MyClient
    from fedicra_amd.networks import net_factory
    from helpers import loader
    res = []
    for tail in (False, True):
        args = argparse.Namespace(strategy="FedICRA", amp=0, model="unet_lc", cid=1, min_num_clients=4, num_classes=2,
                                  img_class="faz", base_lr=0.01, max_iterations=200, iters=5, rep_iters=2, alpha=1.0,
                                  snapshot_path=None, use_graph=use_graph)
        torch.manual_seed(2022)
        ops.manual_seed(11)
        net = net_factory(args, net_type="unet_lc", in_chns=1, class_num=2).to(DEV)
        batches = loader(3, 4, 64, cid=1, device=DEV)
        client = MyClient(args, MyModel(args, net, batches, batches), batches, batches)
        assert client.probe_beside
        client.probe_tail_beside = tail
        cfg = {"iter_global": 60, "iters": 5, "eval_iters": 10, "batch_size": 4, "stage": "fit"}
        client._train(cfg)
        client._train(cfg)
        torch.cuda.synchronize()
        res.append((list(client.last_losses), net.flat_state.clone(), net.flat_counters.clone()))
    (l0, s0, c0), (l1, s1, c1) = res
    assert torch.equal(c0, c1)
    assert np.allclose(l0, l1, rtol=0, atol=2e-5), (l0, l1)
    assert torch.allclose(s0, s1, rtol=1e-4, atol=2e-5), float((s0 - s1).abs().max())
