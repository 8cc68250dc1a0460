"""fedicra_amd/dataloaders/h5mini.py against arrays decoded WITHOUT any HDF5 code (the PNG originals of the reference's
data sets, oracle/gen_h5_fixtures.py): bit-exact on the committed data files, and -- where /root/reference is present --
on a sample of all 3115 files of the reference's data directory."""
import glob
import os
import shutil

import numpy as np
import pytest

from fedicra_amd.dataloaders import h5mini
from fedicra_amd.dataloaders.dataset import BaseDataSets

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _normalise(planes):
    p = planes.astype(np.float32)
    return (p - p.min()) / (p.max() - p.min())


def _fixture():
    return np.load(os.path.join(GOLD, "g21_h5_decoded.npz"))


@pytest.mark.parametrize("tag", ["faz_train_10043", "faz_test_10280", "odoc_train_n0296", "odoc_test_G-22-L"])
def test_fixture_files_decode_bit_exact(tag):
    g = _fixture()
    want = {k.split("/", 1)[1]: g[k] for k in g.files if k.startswith(tag + "/")}
    with h5mini.File(os.path.join(GOLD, "h5", tag + ".h5"), "r") as f:
        assert sorted(f.keys()) == sorted(want)
        assert "image" in f and "no_such" not in f and len(f) == len(want)
        for name, exp in want.items():
            ds = f[name]
            got = ds[:]
            if name == "image":
                exp = _normalise(exp)
                assert ds.dtype == np.float32 and got.dtype == np.float32
            else:
                assert ds.dtype == np.uint8
            assert ds.shape == exp.shape == got.shape
            np.testing.assert_array_equal(got, exp)
            np.testing.assert_array_equal(ds[()], got)
            np.testing.assert_array_equal(ds[...], got)
            np.testing.assert_array_equal(ds[3:7], got[3:7])
        with pytest.raises(KeyError):
            f["no_such"]


def test_unsupported_and_damaged_files_raise(tmp_path):
    src = os.path.join(GOLD, "h5", "faz_test_10280.h5")
    raw = open(src, "rb").read()
    bad = tmp_path / "bad.h5"
    bad.write_bytes(b"not hdf5" + raw[8:])
    with pytest.raises(h5mini.H5Error, match="signature"):
        h5mini.File(str(bad))
    bad.write_bytes(raw[:8] + b"\x02" + raw[9:])                   # a version-2 superblock
    with pytest.raises(h5mini.H5Error, match="superblock version 2"):
        h5mini.File(str(bad))
    bad.write_bytes(raw[:13] + b"\x04\x04" + raw[15:])             # 4-byte offsets
    with pytest.raises(h5mini.H5Error, match="offsets"):
        h5mini.File(str(bad))
    bad.write_bytes(raw[:len(raw) // 2])                           # truncated: a chunk lies past the end
    with pytest.raises(h5mini.H5Error):
        f = h5mini.File(str(bad))
        for k in f.keys():
            f[k][:]
    with pytest.raises(h5mini.H5Error, match="only reads"):
        h5mini.File(src, "w")
    # a damaged deflate stream must not decode silently
    with h5mini.File(src) as f:
        ds = f["image"]
        addr = ds._layout[1]
    hit = raw.index(b"\x78", addr + 200)                           # the first zlib header after the chunk B-tree node
    bad.write_bytes(raw[:hit + 8] + bytes(64) + raw[hit + 72:])
    with pytest.raises(h5mini.H5Error):
        h5mini.File(str(bad))["image"][:]


def test_damaged_files_raise_h5error_only_and_never_hang(tmp_path):
    """every byte read of the decoder is bounds-checked and B-tree / continuation walks are bounded, so a damaged
    file ends in H5Error (or decodes) -- never IndexError / ValueError from a raw index, never an endless loop: a chunk
    B-tree node that lists ITSELF as its child, and 300 files with 24 random bytes overwritten in the metadata region."""
    src = os.path.join(GOLD, "h5", "faz_test_10280.h5")
    raw = bytearray(open(src, "rb").read())
    bad = tmp_path / "bad.h5"
    with h5mini.File(src) as f:
        root = f["image"]._layout[1]
        rank = len(f["image"].shape)
    node = bytearray(raw)
    node[root + 5] = 1                                              # an interior node ...
    node[root + 6:root + 8] = (1).to_bytes(2, "little")             # ... with one entry ...
    key = 8 + 8 * (rank + 1)
    node[root + 24 + key:root + 24 + key + 8] = int(root).to_bytes(8, "little")     # ... whose first child is itself
    bad.write_bytes(bytes(node))
    with pytest.raises(h5mini.H5Error, match="revisits"):
        h5mini.File(str(bad))["image"][:]
    rng = np.random.default_rng(7)
    decoded = failed = 0
    for trial in range(300):
        b = bytearray(raw)
        lim = min(len(b), 6000)                                     # superblock, group tree, heap, object headers live here
        for pos in rng.integers(8, lim, 24):
            b[pos] = int(rng.integers(0, 256))
        bad.write_bytes(bytes(b))
        try:
            with h5mini.File(str(bad)) as f:
                for k in f.keys():
                    f[k][:]
            decoded += 1
        except (h5mini.H5Error, KeyError):
            failed += 1
    assert decoded + failed == 300 and failed > 0


def test_dataset_decodes_once_and_hands_out_copies():
    src = os.path.join(GOLD, "h5", "faz_test_10280.h5")
    with h5mini.File(src) as f:
        ds = f["image"]
        a = ds[:]
        a[...] = 0
        b = ds[:]
        assert ds._cache is not None and b.any() and b is not ds._cache
        np.testing.assert_array_equal(ds[2:5], b[2:5])


def test_base_datasets_reads_the_reference_layout(tmp_path, monkeypatch):
    """BaseDataSets(base_dir, split, client, sup_type) over a DomainN/{train,test} tree of the committed files
    (/root/reference/code/dataloaders/dataset.py:63-96: train labels from sup_type, validation labels from 'mask')."""
    import builtins
    real_import = builtins.__import__

    def no_h5py(name, *a, **k):
        if name == "h5py":
            raise ImportError("h5py is not part of this image")
        return real_import(name, *a, **k)

    monkeypatch.setattr(builtins, "__import__", no_h5py)
    for d in range(1, 6):
        os.makedirs(tmp_path / "Domain{}".format(d) / "train")
        os.makedirs(tmp_path / "Domain{}".format(d) / "test")
    shutil.copy(os.path.join(GOLD, "h5", "faz_train_10043.h5"), tmp_path / "Domain2" / "train" / "10043.h5")
    shutil.copy(os.path.join(GOLD, "h5", "faz_test_10280.h5"), tmp_path / "Domain2" / "test" / "10280.h5")
    g = _fixture()
    tr = BaseDataSets(str(tmp_path), split="train", client="client2", sup_type="scribble", img_class="faz")
    assert len(tr) == 1 and tr.sample_list == ["Domain2/train/10043.h5"]
    np.testing.assert_array_equal(tr[0]["image"], _normalise(g["faz_train_10043/image"]))
    np.testing.assert_array_equal(tr[0]["label"], g["faz_train_10043/scribble"])
    va = BaseDataSets(str(tmp_path), split="val", client="client2", sup_type="scribble", img_class="faz")
    np.testing.assert_array_equal(va[0]["label"], g["faz_test_10280/mask"])
    assert va[0]["idx"] == 0
    with pytest.raises(KeyError):
        BaseDataSets(str(tmp_path), split="train", client="client2", sup_type="no_such_label", img_class="faz")
    assert len(BaseDataSets(str(tmp_path), split="train", client="client1", sup_type="scribble", img_class="faz")) == 0


@pytest.mark.skipif(not os.path.isdir("/root/reference/data"), reason="the reference's data directory is not on this box")
def test_sweep_of_the_reference_data_directory():
    """Every 29th of the reference's .h5 files, every data set in it, against the PNG it was made from."""
    pytest.importorskip("PIL")
    from oracle.gen_h5_fixtures import png_arrays
    files = sorted(glob.glob("/root/reference/data/*_h5/Domain*/*/*.h5"))
    assert len(files) > 3000
    checked = 0
    for path in files[::29]:
        dataset = path.split("/")[-4][:-3]
        domain, split, stem = path.split("/")[-3], path.split("/")[-2], os.path.basename(path)[:-3]
        want = png_arrays(dataset, domain, split, stem)
        with h5mini.File(path, "r") as f:
            assert sorted(f.keys()) == sorted(want), path
            for name, exp in want.items():
                got = f[name][:]
                np.testing.assert_array_equal(got, _normalise(exp) if name == "image" else exp, err_msg=path + ":" + name)
                checked += 1
    assert checked > 300
