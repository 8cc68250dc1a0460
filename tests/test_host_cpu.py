"""CPU-side checks: C-ABI library loads and exports every declared symbol (no compute), module surface /
flat state / wire format, strategy arithmetic vs the oracle, 2-process gloo aggregation."""
import ctypes
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    """A rendezvous port nobody holds right now (a pid-derived constant collided with a socket of the previous test still in
    TIME_WAIT once in a while: one spurious failure in ~20 runs of this file)."""
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def test_library_exports_every_declared_symbol():
    from fedicra_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__ as g
        g.build()
    lib = ctypes.CDLL(_lib.LIB_PATH)
    hdr = open(os.path.join(ROOT, "include", "fedicra_hip.h")).read()
    declared = sorted(set(re.findall(r"^(?:int|long)\s+(fi_\w+)\s*\(", hdr, flags=re.M)))
    assert len(declared) >= 27
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/fedicra_hip.h but not exported"
    assert sorted(_lib.EXPORTS) == declared
    lib.fi_abi_version.restype = ctypes.c_int
    from fedicra_amd import _lib as L
    header = open(os.path.join(ROOT, "include", "fedicra_hip.h")).read()
    assert int(re.search(r"#define FI_ABI_VERSION (\d+)", header).group(1)) == L.ABI_VERSION
    assert lib.fi_abi_version() == L.ABI_VERSION      # host-only call: no GPU needed


def test_product_has_no_cpu_fallback_and_no_oracle_import():
    """The product path must fail loudly without a device and must never import the oracle."""
    from fedicra_amd import _lib
    with pytest.raises(_lib.FiError):
        _lib.maxpool2_fwd(torch.zeros(1, 2, 2, 8), torch.zeros(1, 1, 1, 8))
    for dirpath, _, files in os.walk(os.path.join(ROOT, "fedicra_amd")):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle", src, flags=re.M), f"{f} imports the oracle"


def test_module_surface_matches_oracle_and_reference_layout():
    from fedicra_amd.networks.unet import UNet, UNet_Head, UNet_LC, UNet_LC_MultiHead, UNet_MultiHead
    from oracle.unet_ref import RefUNet, RefUNetLC
    pairs = [(lambda: UNet(1, 2), lambda: RefUNet(1, 2), 136), (lambda: UNet_Head(3, 3), lambda: RefUNet(3, 3, 1), 144),
             (lambda: UNet_MultiHead(1, 2), lambda: RefUNet(1, 2, 3), 160),
             (lambda: UNet_LC(1, 2, 1, 8, 8, 3), lambda: RefUNetLC(1, 2, 1, 8, 8, 3, 1), 144),
             (lambda: UNet_LC_MultiHead(1, 2, 1, 8, 8, 3), lambda: RefUNetLC(1, 2, 1, 8, 8, 3, 3), 160)]
    for mk, mkref, n in pairs:
        torch.manual_seed(2022)
        m = mk()
        torch.manual_seed(2022)
        r = mkref()
        sm, sr = m.state_dict(), r.state_dict()
        assert list(sm.keys()) == list(sr.keys()) and len(sm) == n
        for k in sm:
            assert sm[k].shape == sr[k].shape and sm[k].dtype == sr[k].dtype, k
            assert torch.equal(sm[k], sr[k]), f"default init differs at {k}"     # same seed -> same init
    torch.manual_seed(2022)
    m = UNet_LC(1, 2, 1, 8, 8, 3)
    assert not any("pcs" in k for k in m.state_dict())             # quirk 1 reproduced
    assert sum(p.numel() for p in m.encoder.pcs_list[0].parameters()) == 8 * 256 + 256 * 256 + 512 * 16 + 16 * 256


def test_flat_state_views_and_wire_format():
    import copy
    from fedicra_amd.flower_common import MyModel
    from fedicra_amd.networks.unet import UNet
    from oracle import fed_ref
    from oracle.unet_ref import RefUNet, seeded_state
    import argparse
    m = UNet(1, 2)
    seeded_state(m, 7)
    r = RefUNet(1, 2)
    seeded_state(r, 7)
    flat = m.flat_state
    w = m.encoder.down1.maxpool_conv[1].conv_conv[0].weight
    assert w.shape == (32, 16, 3, 3) and w.permute(0, 2, 3, 1).is_contiguous()      # [Cout][kh][kw][Cin] memory
    assert flat.data_ptr() <= w.data_ptr() < flat.data_ptr() + flat.numel() * 4
    rm = m.encoder.in_conv.conv_conv[1].running_mean
    assert flat.data_ptr() <= rm.data_ptr() < flat.data_ptr() + flat.numel() * 4
    args = argparse.Namespace(strategy="FedAvg", amp=0, cid=0, num_classes=2, img_class="faz")
    mm = MyModel(args, m, [], [])
    ws, wr = [a.copy() for a in mm.get_weights(None)], fed_ref.get_weights(r)   # .numpy() aliases CPU storage
    assert len(ws) == 136 and sum(a.size for a in ws if a.dtype == np.float32) == 1816418
    for a, b in zip(ws, wr):
        assert a.shape == b.shape and a.dtype == b.dtype and np.array_equal(a, b)
    # set_weights (plain) incl. the int64 <- float64 truncation (quirk 6)
    ws2 = [a * 2 if a.dtype == np.float32 else np.float64(7.9) for a in ws]
    mm.set_weights(ws2, {})
    assert int(m.encoder.in_conv.conv_conv[1].num_batches_tracked) == 7
    assert torch.equal(m.flat_counters, torch.full((18,), 7, dtype=torch.int64))
    np.testing.assert_array_equal(mm.get_weights(None)[0], ws[0] * 2)
    m2 = copy.deepcopy(m)
    assert m2.flat_state.data_ptr() != flat.data_ptr() and torch.equal(m2.flat_state, m.flat_state)
    names = [n for n, _ in m.named_parameters()]
    body = [n for n in names if "out_conv" not in n]
    assert len(m.param_ranges(names)) == 1 and len(m.param_ranges(["decoder.out_conv.weight", "decoder.out_conv.bias"])) == 1
    assert len(m.param_ranges(body)) == 1                       # out_conv is the last parameter of UNet
    dec = [n for n in names if any(k in n for k in ("out_conv", "up4", "up3", "up2", "up1"))]
    assert len(dec) == 42 and len(m.param_ranges(dec)) == 1    # ALA range is contiguous


def test_host_aggregate_matches_oracle_and_fl_roundtrip():
    from fedicra_amd import fl
    from fedicra_amd.flower_common import FedAvg, FedICRA, aggregate, fit_metrics_aggregation_fn, get_strategy
    from oracle import fed_ref
    rng = np.random.default_rng(1)
    n = [21, 13, 17, 59, 3]
    ws = [[rng.normal(size=(4, 3, 3, 3)).astype(np.float32), rng.normal(size=(4,)).astype(np.float32),
           np.array(5 + k, dtype=np.int64)] for k in range(5)]
    a, b = aggregate(list(zip(ws, n))), fed_ref.fedavg_aggregate(list(zip(ws, n)))
    for x, y in zip(a, b):
        assert x.dtype == y.dtype and np.array_equal(x, y)
    p = fl.ndarrays_to_parameters(ws[0])
    back = fl.parameters_to_ndarrays(p)
    assert all(np.array_equal(x, y) and x.dtype == y.dtype for x, y in zip(back, ws[0]))
    strat = get_strategy("FedICRA", fit_metrics_aggregation_fn=fit_metrics_aggregation_fn, accept_failures=False)
    assert isinstance(strat, FedICRA) and isinstance(strat, FedAvg) and repr(strat) == "FedICRA(accept_failures=False)"
    res = [(None, fl.FitRes(fl.Status("OK", "Success"), fl.ndarrays_to_parameters(w), k, {f"client_{i}_lr": 0.1}))
           for i, (w, k) in enumerate(zip(ws, n))]
    params, metrics = strat.aggregate_fit(10, res, [])
    for x, y in zip(fl.parameters_to_ndarrays(params), b):
        assert np.array_equal(x, y)
    assert set(metrics) == {f"client_{i}_lr" for i in range(5)}
    assert strat.aggregate_fit(10, res, [RuntimeError()]) == (None, {})      # accept_failures=False


def test_evaluate_metrics_aggregation_fn():
    import argparse
    from fedicra_amd.flower_common import VAL_METRICS, get_evaluate_metrics_aggregation_fn
    args = argparse.Namespace(min_num_clients=2, num_classes=2)
    fn = get_evaluate_metrics_aggregation_fn(args, VAL_METRICS)
    em = []
    for c, (n, d) in enumerate([(10, 0.5), (30, 0.9)]):
        m = {}
        for name in VAL_METRICS:
            m[f"client_{c}_val_1_{name}"] = d
            m[f"client_{c}_val_mean_{name}"] = d
        em.append((n, m))
    out = fn(em)
    assert abs(out["val_mean_dice"] - (10 * 0.5 + 30 * 0.9) / 40) < 1e-12
    assert abs(out["val_avg_mean_dice"] - 0.7) < 1e-12 and abs(out["val_1_jc"] - 0.8) < 1e-12


def test_metrics_from_counts_against_oracle_dice():
    from fedicra_amd.flower_common import metrics_from_counts
    from oracle.losses_ref import dice_percase
    rng = np.random.default_rng(3)
    for _ in range(20):
        p, g = rng.random((16, 16)) < 0.3, rng.random((16, 16)) < 0.4
        tp, npred, ngt = int((p & g).sum()), int(p.sum()), int(g.sum())
        m = metrics_from_counts(tp, npred, ngt, 256)
        assert m[0] == dice_percase(p.copy(), g.copy())
    assert metrics_from_counts(0, 0, 5, 256) == [0.0] * 7          # empty prediction (val_2D.py:21-22)


_WORKER = r'''
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
rank, world = int(sys.argv[2]), int(sys.argv[3])
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=sys.argv[4], RANK=str(rank), WORLD_SIZE=str(world))
dist.init_process_group("gloo", rank=rank, world_size=world)
from fedicra_amd.comm import WeightedAllReduce
from fedicra_amd.flower_common import DeviceWeights
n = [21, 13, 17][:world]
g = torch.Generator().manual_seed(100 + rank)
state = torch.randn(1000, generator=g)
cnt = torch.tensor([10 * (rank + 1) + 3, 7], dtype=torch.int64)
agg = WeightedAllReduce(n[rank], device=None)
out = agg.aggregate(DeviceWeights(state, cnt))
states = [torch.randn(1000, generator=torch.Generator().manual_seed(100 + r)) for r in range(world)]
ref = sum(s.double() * k for s, k in zip(states, n)) / sum(n)
assert agg.all_n == n and agg.total == sum(n)
assert torch.allclose(out.state.double(), ref, atol=1e-6), (out.state.double() - ref).abs().max()
ci = [int(sum((10 * (r + 1) + 3) * n[r] for r in range(world)) / sum(n)), 7]
assert out.counters.tolist() == ci, (out.counters.tolist(), ci)
dist.barrier(); dist.destroy_process_group()
print("rank", rank, "ok")
'''


@pytest.mark.timeout(300)
@pytest.mark.parametrize("world", [2, 3])
def test_weighted_allreduce_gloo_multiprocess(tmp_path, world):
    script = tmp_path / "w.py"
    script.write_text(_WORKER)
    port = str(_free_port())
    procs = [subprocess.Popen([sys.executable, str(script), ROOT, str(r), str(world), port], stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT) for r in range(world)]
    from helpers import communicate_all
    outs = communicate_all(procs, timeout=240)
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o


def test_unet3d_surface_keys_and_init_match_reference_golden(golden):
    """a18 module surface: same state_dict keys as the reference's unet_3D and, under the same torch seed, the same
    initial state (the constructors consume the RNG in the reference's order)."""
    from fedicra_amd.networks.net_factory_3d import net_factory_3d
    from fedicra_amd.networks.unet_3D import unet_3D
    from helpers import assert_ck
    g = golden("g9_unet3d.npz")
    torch.manual_seed(11)
    m = unet_3D(n_classes=2, in_channels=1)
    assert list(m.state_dict().keys()) == [str(k) for k in g["keys"]]
    for k, v in m.state_dict().items():
        assert_ck(v.double(), g["init_seed11/" + k], what=k)
    with pytest.raises(NotImplementedError):
        net_factory_3d("voxresnet")


def test_c_abi_rejects_bad_arguments_without_touching_the_gpu():
    """Argument validation happens before any launch: NULL pointers, empty / ragged shapes, channel counts the vector
    kernels cannot serve and unknown dtypes come back as FI_ERR_* codes (nothing is thrown across the ABI, nothing is
    launched -- so this runs on the CPU-only box too)."""
    lib = ctypes.CDLL(os.path.join(ROOT, "fedicra_amd", "libfedicra_hip.so"))

    class FiConv(ctypes.Structure):
        _fields_ = [(n, ctypes.c_int) for n in ("dtype", "N", "H", "W", "ksize", "c0", "c1", "co0", "co1", "accumulate0",
                                                "accumulate1", "y_f32")]

    class FiBnAct(ctypes.Structure):
        _fields_ = [("dtype", ctypes.c_int), ("pixels", ctypes.c_long), ("C", ctypes.c_int), ("hw", ctypes.c_int),
                    ("slope", ctypes.c_float), ("drop_mode", ctypes.c_int), ("drop_p", ctypes.c_float),
                    ("seed", ctypes.c_uint64), ("mask", ctypes.c_void_p), ("seed_offset", ctypes.c_void_p)]
    ERR_DTYPE, ERR_SHAPE, ERR_UNSUPPORTED, ERR_NULL = -1, -2, -3, -4
    p = ctypes.c_void_p(0x1000)                      # never dereferenced: every call below fails validation first
    ok = FiConv(0, 1, 8, 8, 3, 16, 0, 16, 0, 0, 0, 0)
    assert lib.fi_conv2d_fwd(ctypes.byref(ok), None, None, p, None, p, None, None, None) == ERR_NULL
    assert lib.fi_conv2d_fwd(ctypes.byref(FiConv(7, 1, 8, 8, 3, 16, 0, 16, 0, 0, 0, 0)), p, None, p, None, p, None, None,
                             None) == ERR_DTYPE
    assert lib.fi_conv2d_fwd(ctypes.byref(FiConv(0, 1, 8, 8, 5, 16, 0, 16, 0, 0, 0, 0)), p, None, p, None, p, None, None,
                             None) == ERR_UNSUPPORTED                                     # 5x5 kernels
    for bad in (FiConv(0, 0, 8, 8, 3, 16, 0, 16, 0, 0, 0, 0), FiConv(0, 1, 0, 8, 3, 16, 0, 16, 0, 0, 0, 0),
                FiConv(0, 1, 8, 8, 3, 0, 0, 16, 0, 0, 0, 0), FiConv(0, 1, 8, 8, 3, 16, 0, 0, 0, 0, 0, 0)):
        assert lib.fi_conv2d_fwd(ctypes.byref(bad), p, None, p, None, p, None, None, None) == ERR_SHAPE   # empty inputs
    assert lib.fi_conv2d_fwd(ctypes.byref(FiConv(0, 1, 8, 8, 3, 16, 8, 16, 0, 0, 0, 0)), p, None, p, None, p, None, None,
                             None) == ERR_NULL                                            # c1 > 0 without x1
    lib.fi_conv2d_wgrad_workspace.restype = ctypes.c_long
    assert lib.fi_conv2d_wgrad_workspace(ctypes.byref(ok)) > 0
    assert lib.fi_conv2d_wgrad(ctypes.byref(ok), p, None, p, p, None, p, ctypes.c_long(16), None) == ERR_SHAPE  # workspace too small
    assert lib.fi_maxpool2_fwd(0, p, p, 1, 7, 8, 16, None) == ERR_SHAPE                    # odd height
    assert lib.fi_maxpool3d_fwd(0, p, p, 1, 4, 4, 5, 8, None) == ERR_SHAPE
    assert lib.fi_maxpool2_fwd(0, p, p, 1, 8, 8, 6, None) == ERR_SHAPE                     # C not a whole 16-byte vector
    assert lib.fi_upsample2x_fwd(1, p, p, 1, 4, 4, 12, None) == ERR_SHAPE
    assert lib.fi_upsample3d2x_fwd(3, p, p, 1, 2, 2, 2, 8, None) == ERR_DTYPE
    bn = FiBnAct(1, 64, 24, 64, 0.01, 0, 0.0, 0, None, None)                               # 24 channels: 3 vectors, 256 % 3 != 0
    assert lib.fi_bn_act_fwd(ctypes.byref(bn), p, p, p, p, None) == ERR_SHAPE
    assert lib.fi_bn_act_bwd_reduce(ctypes.byref(bn), p, p, p, p, p, p, None, None) == ERR_NULL
    assert lib.fi_ce_fwd(p, p, ctypes.c_long(64), 9, 9, p, None) == ERR_SHAPE              # > 8 classes
    assert lib.fi_ce_fwd(None, p, ctypes.c_long(64), 2, 2, p, None) == ERR_NULL
    assert lib.fi_adamw_step(p, p, p, p, ctypes.c_long(0), p, ctypes.c_float(0.9), ctypes.c_float(0.999),
                             ctypes.c_float(1e-8), None, None) == 0                        # empty range: no-op
    assert lib.fi_wgrad_reduce_multi(None, 3, 10, None) == ERR_NULL
    assert lib.fi_wgrad_reduce_multi(p, 0, 0, None) == 0
    assert lib.fi_wgrad_permute3d_multi(None, 3, 10, None) == ERR_NULL
    assert lib.fi_wgrad_permute3d_multi(p, 2, 0, None) == 0                                # no 3x3x3 rows: no launch
    # 3D entries: validated before the first depth-slice launch
    taps = (ctypes.c_void_p * 3)(0x1000, 0x1000, 0x1000)
    assert lib.fi_conv3d_fwd(ctypes.byref(ok), 4, None, None, taps, None, p, None, ctypes.c_long(0), None) == ERR_NULL
    assert lib.fi_conv3d_fwd(ctypes.byref(ok), 0, p, None, taps, None, p, None, ctypes.c_long(0), None) == ERR_SHAPE
    assert lib.fi_conv3d_fwd(ctypes.byref(FiConv(0, 1, 8, 8, 3, 16, 8, 16, 0, 0, 0, 0)), 4, p, None, taps, None, p, None,
                             ctypes.c_long(0), None) == ERR_SHAPE                          # c1 > 0 without x1
    assert lib.fi_conv3d_dgrad(ctypes.byref(FiConv(0, 1, 8, 8, 3, 16, 0, 8, 8, 1, 1, 0)), 4, p, taps, p, None, None) == ERR_SHAPE
    lib.fi_conv3d_wgrad_workspace.restype = ctypes.c_long
    assert lib.fi_conv3d_wgrad_workspace(ctypes.byref(ok), 4) >= lib.fi_conv2d_wgrad_workspace(ctypes.byref(ok))
    assert lib.fi_conv3d_wgrad_workspace(ctypes.byref(ok), 0) == ERR_SHAPE
    assert lib.fi_conv3d_wgrad(ctypes.byref(ok), 4, p, None, p, None, None, p, ctypes.c_long(1 << 20), None) == ERR_NULL
    # ConvTranspose(k 2, s 2) composites
    assert lib.fi_convtranspose2x_fwd(0, 1, 1, 8, 8, 16, 16, 0, None, p, None, p, p, None) == ERR_NULL
    assert lib.fi_convtranspose2x_fwd(0, 1, 2, 8, 8, 16, 16, 0, p, p, None, p, p, None) == ERR_SHAPE   # 2D with depth > 1
    assert lib.fi_convtranspose2x_dgrad(0, 1, 1, 0, 8, 16, 16, 0, p, p, p, p, None) == ERR_SHAPE
    lib.fi_convtranspose2x_wgrad_workspace.restype = ctypes.c_long
    assert lib.fi_convtranspose2x_wgrad_workspace(0, 2, 4, 8, 8, 16, 8, 1) > 0
    assert lib.fi_convtranspose2x_wgrad_workspace(0, 2, 4, 8, 8, 0, 8, 1) == ERR_SHAPE
    assert lib.fi_convtranspose2x_wgrad(0, 1, 1, 8, 8, 16, 16, 0, p, p, None, None, p, ctypes.c_long(64), None) == ERR_NULL


def test_header_is_plain_c_and_links_from_a_c_host(tmp_path):
    """include/fedicra_hip.h compiles as C99 (-pedantic: no C++-isms cross the boundary) and a C
    host links libfedicra_hip.so directly; the calls made here are host-side planning / validation only (no GPU)."""
    import shutil
    import subprocess
    if shutil.which("gcc") is None:
        pytest.skip("gcc not available")
    src = tmp_path / "host.c"
    src.write_text(r'''
#include <stdio.h>
#include "fedicra_hip.h"
int main(void) {
  FiConv d = { FI_BF16, 12, 256, 256, 3, 16, 0, 16, 0, 0, 0, 0, NULL, 0 };   /* no chunk-major second operand */
  long ws = fi_conv2d_wgrad_workspace(&d);
  long ws3 = fi_conv3d_wgrad_workspace(&d, 8);
  int rc = fi_conv2d_fwd(&d, NULL, NULL, NULL, NULL, NULL, NULL, NULL, NULL);
  printf("%ld %ld %d\n", ws, ws3, rc);
  return ws > 0 && ws3 >= ws && rc == FI_ERR_NULL ? 0 : 1;
}
''')
    exe = tmp_path / "host"
    libdir = os.path.join(ROOT, "fedicra_amd")
    cc = subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", os.path.join(ROOT, "include"),
                         str(src), "-o", str(exe), "-L", libdir, "-lfedicra_hip", "-Wl,-rpath," + libdir,
                         "-Wl,-rpath,/opt/rocm/lib"], capture_output=True, text=True)
    assert cc.returncode == 0, cc.stderr
    run = subprocess.run([str(exe)], capture_output=True, text=True, timeout=60)
    assert run.returncode == 0, (run.stdout, run.stderr)


def test_metric_aggregation_matches_reference_golden(golden):
    """Strategy protocol: get_evaluate_metrics_aggregation_fn / fit_metrics_aggregation_fn give the same
    keys, in the same order, with the same values as the reference's own functions (g14)."""
    import argparse
    from fedicra_amd.flower_common import fit_metrics_aggregation_fn, get_evaluate_metrics_aggregation_fn
    g = golden("g14_metric_aggregation.npz")
    counts = [int(n) for n in g["counts"]]
    names = [str(n) for n in g["names"]]
    em = [(counts[c], dict(zip(map(str, g[f"in{c}_keys"]), map(float, g[f"in{c}_vals"])))) for c in range(3)]
    fm = [(counts[c], dict(zip(map(str, g[f"fit{c}_keys"]), map(float, g[f"fit{c}_vals"])))) for c in range(3)]
    out = get_evaluate_metrics_aggregation_fn(argparse.Namespace(min_num_clients=3, num_classes=3), names)(em)
    assert list(out.keys()) == [str(k) for k in g["out_keys"]]
    np.testing.assert_allclose(np.array([float(v) for v in out.values()]), g["out_vals"], rtol=1e-15, atol=0)
    fout = fit_metrics_aggregation_fn(fm)
    assert list(fout.keys()) == [str(k) for k in g["fit_out_keys"]]
    np.testing.assert_array_equal(np.array([float(v) for v in fout.values()]), g["fit_out_vals"])


def test_run_federated_host_logic_follows_the_launcher_and_server_loop(tmp_path):
    """fedicra_amd.run_federated without a GPU: the launcher's flags and defaults (flower_runner.py:18-55), the client
    script's argument checks (flower_pCE_2D.py:262-276), the round schedule and config dicts of MyServer.fit / fit_config
    (flower_common.py:256, flower_pCE_2D.py:320-338) and the checkpoint names (flower_common.py:343-365)."""
    import pytest
    from fedicra_amd import run_federated as rf
    p = rf.build_parser()
    a = p.parse_args(["--exp", "e1"])
    assert (a.procedure, a.base_lr, a.model, a.img_class, a.max_iterations, a.iters, a.eval_iters, a.alpha, a.batch_size,
            a.tree_loss_weight, a.strategy, a.img_size, a.amp, a.rep_iters) == \
        ("flower_pCE_2D", 0.01, "unet", "faz", 30000, 10, 20, 0.5, 12, 0.1, "FedAvg", 256, 0, 3)
    a.snapshot_dir = str(tmp_path)
    sup = rf.check_args(a, 5)
    assert sup == ["scribble_noisy", "keypoint", "block", "box", "scribble"] and (a.num_classes, a.in_chns) == (2, 1)
    assert a.min_num_clients == 5 and a.root_path == "../data/FAZ_h5" and a.snapshot_path.endswith("e1")
    assert list(rf.round_schedule(a))[:3] == [10, 20, 30] and list(rf.round_schedule(a))[-1] == 30000
    assert rf.make_config(a, 20, "fit") == {"iter_global": 20, "iters": 10, "eval_iters": 20, "batch_size": 12, "stage": "fit"}
    import os
    b = os.path.basename
    assert [b(n) for n in rf.checkpoint_names(a, 40, 0.87654321)] == ["iter_40_dice_0.8765.pth", "unet_best_model.pth"]
    assert [b(n) for n in rf.checkpoint_names(a, 40, client_id=3, client_dice=0.5)] == \
        ["client_3_iter_40_dice_0.5.pth", "client_3_unet_best_model.pth"]
    assert [b(n) for n in rf.checkpoint_names(a, 3000)] == ["iter_3000.pth"]
    assert [b(n) for n in rf.checkpoint_names(a, 3000, client_id=1)] == ["client_1_iter_3000.pth"]
    bad = p.parse_args(["--exp", "e", "--eval_iters", "15"])
    with pytest.raises(AssertionError):
        rf.check_args(bad, 2)                                    # eval_iters must be a multiple of iters
    icra = p.parse_args(["--exp", "e", "--strategy", "FedICRA", "--model", "unet"])
    with pytest.raises(AssertionError):
        rf.check_args(icra, 2)                                   # FedICRA needs the LC model
    odoc = p.parse_args(["--exp", "e", "--img_class", "odoc"])
    assert rf.check_args(odoc, 5)[3] == "keypoint" and (odoc.num_classes, odoc.in_chns) == (3, 3)
    typo = p.parse_args(["--exp", "e", "--strategy", "Fedicra"])
    with pytest.raises(AssertionError, match="unknown --strategy"):
        rf.check_args(typo, 2)                                   # get_strategy asserts the name (flower_common.py:431-433)
    for name in ("FedAdagrad", "FedAdam", "FedYogi"):
        rf.check_args(p.parse_args(["--exp", "e", "--strategy", name]), 2)
    assert p.parse_args(["--exp", "e"]).adamw_frozen == "torch2"


def _const_term_worker(rank, world, port, q):
    import os
    import torch
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from fedicra_amd.comm import WeightedAllReduce
    from fedicra_amd.flower_common import DeviceWeights
    n_k = [3, 5][rank]
    state = torch.arange(6, dtype=torch.float32) + 10 * rank
    cnt = torch.tensor([rank + 1, 7], dtype=torch.int64)
    absent = (DeviceWeights(torch.full((6,), 100.0), torch.tensor([4, 4], dtype=torch.int64)), 2)
    agg = WeightedAllReduce(n_k, device=None, constant_term=absent)
    out = agg.aggregate(DeviceWeights(state, cnt))
    q.put((rank, out.state.clone(), out.counters.clone(), agg.total))
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_weighted_allreduce_constant_term_counts_the_clients_no_rank_hosts():
    """bench.py with fewer GPUs than the federation has clients: the absent clients' n_k * state enters the weighted sum
    once (rank 0 adds it), the total weight includes them, every rank gets the same mean (world_size 2, gloo)."""
    import torch
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_const_term_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    try:
        res = sorted([q.get(timeout=120) for _ in ps], key=lambda t: t[0])
    finally:
        for p in ps:
            p.join(timeout=60)
            if p.is_alive():                                 # a rank stuck in the rendezvous must not outlive the test
                p.kill()
    base = torch.arange(6, dtype=torch.float32)
    want = (3 * base + 5 * (base + 10) + 2 * 100.0) / 10
    want_c = ((torch.tensor([1, 7]) * 3 + torch.tensor([2, 7]) * 5 + torch.tensor([4, 4]) * 2).double() / 10).to(torch.int64)
    for rank, st, cn, total in res:
        assert total == 10 and torch.allclose(st, want) and torch.equal(cn, want_c), (rank, st, cn)


def test_g8_metrics_from_counts_reproduce_the_references_rows(golden):
    """The product's count-based metric rows (flower_common.metrics_from_counts) against golden g8 = the reference's own
    calculate_metric_percase / test_single_volume rows: class rule, empty-prediction rule, all seven columns."""
    from fedicra_amd.flower_common import metrics_from_counts
    from oracle.losses_ref import hd95_percase
    g = golden("g8_eval_metrics.npz")
    for name, ncls in (("faz", 2), ("odoc", 3)):
        for p, q, want in zip(g[f"{name}/pred"], g[f"{name}/gt"], g[f"{name}/per_image"]):
            for k in range(1, ncls):
                P = (p == 1) if k == 1 else (p >= 1)
                G = (q == 1) if k == 1 else (q >= 1)
                row = metrics_from_counts(int((P & G).sum()), int(P.sum()), int(G.sum()), P.size, hd95_percase(P, G))
                np.testing.assert_allclose(np.array(row), want[k - 1], rtol=0, atol=1e-15, err_msg=f"{name} class {k}")


def test_g6_host_aggregate_and_set_weights_follow_the_references_round_trip(golden):
    """The Flower-compatible host path of the product (get_weights -> aggregate -> set_weights on lists of numpy arrays)
    against golden g6: aggregate dtypes, float64 counters, truncating load, loaded tensors."""
    import argparse
    from fedicra_amd.flower_common import MyModel, aggregate
    from fedicra_amd.networks.unet import UNet
    from oracle.unet_ref import seeded_state
    g = golden("g6_fedavg_counters.npz")
    n_all = [int(v) for v in g["n_all"]]
    keys = [str(k) for k in g["keys"]]
    args = argparse.Namespace(strategy="FedAvg", amp=0, cid=0, num_classes=2, img_class="faz")
    for K in (2, 5):
        results = []
        for k in range(K):
            net = UNet(1, 2)
            seeded_state(net, 100 + k)
            for j in range(net.flat_counters.numel()):
                net.flat_counters[j] = 7 * k + 3 * j + 1
            results.append(([a.copy() for a in MyModel(args, net, [], []).get_weights(None)], n_all[k]))
        assert list(net.state_dict().keys()) == keys
        agg = aggregate(results)
        assert [str(a.dtype) for a in agg] == [str(s) for s in g[f"K{K}/dtypes"]]
        cidx = [i for i, k_ in enumerate(keys) if k_.endswith("num_batches_tracked")]
        np.testing.assert_array_equal(np.array([float(agg[i]) for i in cidx]), g[f"K{K}/counters_f64"])
        recv = UNet(1, 2)
        seeded_state(recv, 7)
        MyModel(args, recv, [], []).set_weights(agg, {"iter_global": 60})
        np.testing.assert_array_equal(recv.flat_counters.numpy(), g[f"K{K}/counters_loaded"])
        np.testing.assert_array_equal(recv.state_dict()["decoder.out_conv.weight"].numpy(), g[f"K{K}/out_conv_weight"])


def test_batch_stager_ring_never_serves_a_batch_from_a_recycled_or_aliased_pair(monkeypatch):
    """staging.BatchStager's slot bookkeeping, driven on the host with stand-ins for the HIP stream / event objects
    (ADVICE r3): a batch prefetched more than a ring length before its use, out-of-order consumption, serial fetches that
    walk over pending pairs, and a dropped host batch whose id() is re-used by a new dict -- the consumer must always see
    the pixels of the batch it asked for."""
    import contextlib
    import gc
    from fedicra_amd import staging

    class Ev:
        def record(self, stream=None): pass
        def query(self): return True

    class St:
        def wait_event(self, ev): pass

    class Hip:
        Stream = staticmethod(lambda device: St())
        Event = staticmethod(lambda: Ev())
        current_stream = staticmethod(lambda: St())
        stream = staticmethod(lambda s: contextlib.nullcontext())
        pinned = staticmethod(lambda t: True)

    monkeypatch.setattr(staging, "_Hip", Hip)
    mk = lambda v: {"image": torch.full((2, 4, 4), float(v)), "label": torch.full((2, 4, 4), v % 250, dtype=torch.uint8)}

    def take(st, b, v):
        x, y = st.fetch(b)
        assert float(x.min()) == float(x.max()) == float(v) and int(y[0, 0, 0]) == v % 250, (v, float(x[0, 0, 0]))
        st.release()

    # (1) trigger 1 of the finding: batch k prefetched, then >= ring-length other batches staged before k is used
    for slots in (2, 3, 4, 12):
        st = staging.BatchStager("cpu", slots=slots)
        late = mk(1000)
        st.prefetch(late)
        others = [mk(i) for i in range(3 * slots + 1)]
        for i, b in enumerate(others):
            if i + 1 < len(others):
                st.prefetch(others[i + 1])
            take(st, b, i)
        take(st, late, 1000)                                # from its own pair, or serially after its prefetch was given up
    # (2) out-of-order consumption with several copies in flight, ring of 2
    st = staging.BatchStager("cpu", slots=2)
    bs = [mk(i) for i in range(8)]
    for b in bs[:4]:
        st.prefetch(b)
    for i in (2, 0, 3, 1):
        take(st, bs[i], i)
    # (3) a held pair is not handed to the next prefetch before release()
    st = staging.BatchStager("cpu", slots=2)
    a, b, c = mk(1), mk(2), mk(3)
    st.prefetch(a); st.prefetch(b)
    xa, ya = st.fetch(a)
    st.prefetch(c)                                          # both pairs taken (a held, b pending): c waits for its turn or drops b
    assert float(xa[0, 0, 0]) == 1.0
    st.release()
    take(st, b, 2); take(st, c, 3)
    # (4) id() aliasing: a prefetched batch is dropped by its owner; a NEW dict (which CPython is free to give the same
    # id) must not be served the stale pair.  The entry holds its batch, so the id cannot be re-used while it is alive.
    st = staging.BatchStager("cpu", slots=4)
    for rep in range(200):
        old = mk(7)
        st.prefetch(old)
        oid = id(old)
        del old
        gc.collect()
        new = mk(9)
        take(st, new, 9)
        assert oid in st._pending and id(new) != oid
        st._pending.clear()
