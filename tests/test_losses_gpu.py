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
