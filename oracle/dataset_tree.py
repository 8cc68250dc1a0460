"""Test infrastructure (never imported by the product): a synthetic copy of the FAZ / ODOC release's directory layout
(/root/reference/code/dataloaders/dataset.py:98-175 lists Domain1..5/{train,test}) whose '.h5 files' are dict entries --
h5py is not installed here.  oracle/gen_golden.py runs the reference's BaseDataSets over it (g16); the CPU tests rebuild
the identical tree for fedicra_amd.dataloaders.BaseDataSets."""
import os

import numpy as np


def dataset_tree(root, store):
    """The Domain1..5/{train,test}/*.h5 tree of the FAZ / ODOC release (dataset.py:98-175) with dict-backed 'files':
    deterministic contents keyed by the relative path (so a test can rebuild the same tree anywhere)."""
    rels = []
    for dom in range(1, 6):
        for split, n in (("train", 3), ("test", 2)):
            os.makedirs(os.path.join(root, f"Domain{dom}", split), exist_ok=True)
            for i in range(n):
                rel = f"Domain{dom}/{split}/case{i}.h5"
                open(os.path.join(root, rel), "wb").close()
                r = np.random.default_rng(1000 * dom + 10 * i + (split == "test"))
                store[os.path.normpath(os.path.join(root, rel))] = {
                    "image": r.random((16, 16)).astype(np.float32), "mask": r.integers(0, 2, (16, 16)).astype(np.uint8),
                    "scribble": r.integers(0, 3, (16, 16)).astype(np.uint8)}
                rels.append(rel)
    return rels
