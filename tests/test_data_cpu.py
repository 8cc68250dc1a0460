"""Host logic of the data path: RNG consumption, gather tables, directory layout.  CPU only."""
import os
import random
import sys
import types

import numpy as np
import pytest

from oracle import augment_ref as A


def test_random_generator_draws_consume_rng_like_the_reference():
    """Same seeds -> the draws of oracle.augment_ref.draw (itself pinned to the reference's outputs by g11)."""
    from fedicra_amd.dataloaders import RandomGenerator
    for cls_ in ("faz", "odoc", "polyp"):
        random.seed(5)
        np.random.seed(6)
        want = [A.draw(cls_) for _ in range(200)]
        tail = (random.random(), np.random.randint(0, 1 << 30))
        random.seed(5)
        np.random.seed(6)
        gen = RandomGenerator([32, 32], cls_)
        assert [gen.draw() for _ in range(200)] == want
        assert (random.random(), np.random.randint(0, 1 << 30)) == tail
        assert {d[0] for d in want} == {-1, 0, 1, 2, 3} and {d[1] for d in want} == {0, 1}


def test_gather_tables_match_scipys_rotation_parameters():
    from fedicra_amd.dataloaders import RandomGenerator
    draws = [(-1, 0, None), (2, 1, -45), (1, 0, 44), (3, 1, 0), (0, 0, None)]
    ip, dp = RandomGenerator.params([4, 3, 2, 1, 0], draws, 37, 37)
    assert ip.tolist() == [[4, -1, 0, 0], [3, 2, 1, 1], [2, 1, 0, 1], [1, 3, 1, 1], [0, 0, 0, 0]]
    for b, (_, _, ang) in enumerate(draws):
        if ang is None:
            assert not dp[b].any()
        else:
            M, off = A.rot_params(ang, 37, 37)
            np.testing.assert_array_equal(dp[b], [M[0, 0], M[0, 1], M[1, 0], M[1, 1], off[0], off[1]])
    with pytest.raises(ValueError):
        RandomGenerator.params([0], [(1, 0, None)], 40, 28)          # odd quarter turn of a non-square image
    RandomGenerator.params([0], [(2, 0, 10)], 40, 28)


def test_base_datasets_layout_and_errors(tmp_path, monkeypatch):
    """DomainN/{train,test}/*.h5 listing (dataset.py:98-175), train label = sup_type, val label = 'mask' (:86-96)."""
    from fedicra_amd.dataloaders import BaseDataSets
    store = {}
    for d in range(1, 6):
        for split, n in (("train", 3), ("test", 2)):
            os.makedirs(tmp_path / f"Domain{d}" / split)
            for i in range(n):
                p = tmp_path / f"Domain{d}" / split / f"s{i}.h5"
                p.write_bytes(b"")
                store[str(p)] = {"image": np.full((4, 4), d + 0.1 * i), "mask": np.full((4, 4), 1, np.uint8),
                                 "scribble": np.full((4, 4), 2, np.uint8)}

    class File(dict):
        def __init__(self, path, mode):
            super().__init__(store[os.path.normpath(path)])

        def __enter__(self):
            return self

        def __exit__(self, *a):
            return False

    monkeypatch.setitem(sys.modules, "h5py", types.SimpleNamespace(File=File))
    tr = BaseDataSets(str(tmp_path), "train", None, "client2", "scribble", "faz")
    va = BaseDataSets(str(tmp_path), "val", None, "client2", "scribble", "faz")
    assert len(tr) == 3 and len(va) == 2 and all(s.startswith("Domain2/train/") for s in tr.sample_list)
    assert tr[0]["label"].max() == 2 and va[0]["label"].max() == 1 and tr[1]["idx"] == 1
    assert len(BaseDataSets(str(tmp_path), "train", None, "client_all", "scribble", "faz")) == 15
    with pytest.raises(KeyError):
        BaseDataSets(str(tmp_path), "train", None, "client9", "scribble", "faz")
    with pytest.raises(NotImplementedError):
        BaseDataSets(str(tmp_path), "train", None, "client1", "random_walker", "faz")
    monkeypatch.setitem(sys.modules, "h5py", None)                    # import h5py -> ImportError: the bundled decoder
    from fedicra_amd.dataloaders.h5mini import H5Error               # takes over and rejects the empty stand-in files
    with pytest.raises(H5Error, match="signature"):
        BaseDataSets(str(tmp_path), "train", None, "client1", "scribble", "faz")


def test_two_stream_batch_sampler():
    from fedicra_amd.dataloaders import TwoStreamBatchSampler
    np.random.seed(0)
    s = TwoStreamBatchSampler(list(range(10)), list(range(100, 104)), batch_size=4, secondary_batch_size=1)
    batches = list(s)
    assert len(s) == 3 and len(batches) == 3
    assert all(len(b) == 4 and sum(x >= 100 for x in b) == 1 for b in batches)
    assert len({x for b in batches for x in b if x < 100}) == 9


_COLOC_WORKER = r'''
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
rank, world = int(sys.argv[2]), int(sys.argv[3])
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=sys.argv[4], RANK=str(rank), WORLD_SIZE=str(world))
dist.init_process_group("gloo", rank=rank, world_size=world)
from fedicra_amd.comm import WeightedAllReduce
from fedicra_amd.flower_common import DeviceWeights
C = 2                                              # clients hosted by every rank; client id = rank * C + j
n = [21, 13, 17, 59]
mk = lambda cid: torch.randn(1000, generator=torch.Generator().manual_seed(100 + cid))
cnt = lambda cid: torch.tensor([10 * (cid + 1) + 3, 7], dtype=torch.int64)
mine = [rank * C + j for j in range(C)]
agg = WeightedAllReduce([n[c] for c in mine], device=None)
out = agg.aggregate([DeviceWeights(mk(c), cnt(c)) for c in mine])
ref = sum(mk(c).double() * n[c] for c in range(world * C)) / sum(n)
assert agg.total == sum(n) and agg.all_n == [n[0] + n[1], n[2] + n[3]]
assert torch.allclose(out.state.double(), ref, atol=1e-6), (out.state.double() - ref).abs().max()
ci = [int(sum((10 * (c + 1) + 3) * n[c] for c in range(world * C)) / sum(n)), 7]
assert out.counters.tolist() == ci, (out.counters.tolist(), ci)
dist.barrier(); dist.destroy_process_group()
'''


@pytest.mark.timeout(300)
def test_weighted_allreduce_with_colocated_clients_gloo(tmp_path):
    """2 processes x 2 clients each: sum_k n_k w_k over the co-located clients, all-reduce, / sum of all n_k = flwr's
    weighted mean over the 4 clients."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "w.py"
    script.write_text(_COLOC_WORKER)
    from helpers import communicate_all, free_port
    port = str(free_port())
    procs = [subprocess.Popen([sys.executable, str(script), root, str(r), "2", port], stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT) for r in range(2)]
    outs = communicate_all(procs, timeout=240)
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o


def test_fedopt_strategies_follow_the_published_update_rules():
    """get_strategy('FedAdagrad' | 'FedAdam' | 'FedYogi') on host payloads: three rounds against the update rules of
    flwr 1.0.0 written out directly (third-party, absent: parity unpinned), dtype rules included (fp32 arrays stay fp32,
    the int64 counter becomes float64 after the first true divide); flwr's mandatory initial_parameters is mandatory."""
    from fedicra_amd import fl
    from fedicra_amd.flower_common import get_strategy
    rng = np.random.default_rng(0)
    init = [rng.standard_normal((4, 3)).astype(np.float32), np.array(5, dtype=np.int64)]

    class Res:
        def __init__(self, seed, n):
            g = np.random.default_rng(seed)
            self.parameters = fl.ndarrays_to_parameters([g.standard_normal((4, 3)).astype(np.float32),
                                                         np.array(7 + seed, dtype=np.int64)])
            self.num_examples, self.metrics = n, {}

    rules = {"FedAdagrad": (1e-1, 0.0, 0.0, 1e-9, lambda v, d, b2: v + d * d),
             "FedAdam": (1e-1, 0.9, 0.99, 1e-9, lambda v, d, b2: b2 * v + (1 - b2) * (d * d)),
             "FedYogi": (1e-2, 0.9, 0.99, 1e-3, lambda v, d, b2: v - (1.0 - b2) * (d * d) * np.sign(v - d * d))}
    for name, (eta, b1, b2, tau, second) in rules.items():
        st = get_strategy(name, initial_parameters=fl.ndarrays_to_parameters(init))
        cur = [np.array(a) for a in init]
        m = v = None
        for rnd in range(3):
            rs = [Res(2 * rnd, 3), Res(2 * rnd + 1, 5)]
            got, _ = st.aggregate_fit(rnd, [(None, r) for r in rs], [])
            ws = [fl.parameters_to_ndarrays(r.parameters) for r in rs]
            avg = [(ws[0][i] * 3 + ws[1][i] * 5) / 8 for i in range(2)]
            d = [a - c for a, c in zip(avg, cur)]
            m = [np.zeros_like(x) for x in d] if m is None else m
            v = [np.zeros_like(x) for x in d] if v is None else v
            m = [b1 * x + (1 - b1) * y for x, y in zip(m, d)]
            v = [second(x, y, b2) for x, y in zip(v, d)]
            cur = [c + eta * x / (np.sqrt(z) + tau) for c, x, z in zip(cur, m, v)]
            out = fl.parameters_to_ndarrays(got)
            assert out[0].dtype == np.float32 and np.asarray(out[1]).dtype == np.float64
            np.testing.assert_array_equal(out[0], cur[0])
            assert float(out[1]) == float(cur[1])
    with pytest.raises(TypeError):
        get_strategy("FedAdam")


def test_two_stream_batch_sampler_matches_reference_golden(golden):
    """Same index batches as the reference's sampler under the same numpy seed, over two epochs (g15)."""
    from fedicra_amd.dataloaders import TwoStreamBatchSampler
    g = golden("g15_two_stream_sampler.npz")
    for tag in ("a", "b"):
        nprim, sec0, nsec, bs, sbs = (int(v) for v in g[f"{tag}/args"])
        np.random.seed(15)
        s = TwoStreamBatchSampler(list(range(nprim)), list(range(sec0, sec0 + nsec)), bs, sbs)
        assert len(s) == int(g[f"{tag}/len"])
        for e in range(2):
            got = np.array([list(map(int, b)) for b in s])
            np.testing.assert_array_equal(got, g[f"{tag}/epoch{e}"])


def test_base_datasets_match_reference_golden(golden, tmp_path, monkeypatch):
    """fedicra_amd.dataloaders.BaseDataSets over the same dict-backed tree the reference's own class was run on (g16):
    sample lists per client, train label = sup_type / val label = 'mask' (the transform applied by __getitem__ is a
    device launch: tests/test_data_gpu.py)."""
    from fedicra_amd.dataloaders import BaseDataSets
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
    for client in ("client1", "client4", "client_all"):
        for split in ("train", "val"):
            ds = BaseDataSets(str(tmp_path), split, None, client, "scribble", "faz")
            names = sorted(ds.sample_list)
            assert names == [str(n) for n in g[f"{client}/{split}/names"]]
            by = {n: ds[i] for i, n in enumerate(ds.sample_list)}
            np.testing.assert_array_equal(np.stack([np.asarray(by[n]["image"]) for n in names]), g[f"{client}/{split}/images"])
            np.testing.assert_array_equal(np.stack([np.asarray(by[n]["label"]) for n in names]), g[f"{client}/{split}/labels"])
