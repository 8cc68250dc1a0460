"""The reference's data path (/root/reference/code/dataloaders/dataset.py:63-330) for a data set
that is resident in HBM.

The reference keeps every decoded sample in host memory (``BaseDataSets.data_list``, :84-96), augments one numpy image
at a time in 4 DataLoader worker processes (``RandomGenerator``, :231-256: rot90/flip and a nearest-neighbour scipy
rotation that pads the label with the ignore class) and copies every batch host->device.  The largest data set of the
paper is 1332 images of 256^2 (350 MB as fp32): here the decoded set is uploaded ONCE (``BaseDataSets.resident``), an
epoch's shuffling and augmentation parameters are drawn on the host in the reference's order (python ``random`` and
``np.random`` -- a few integers per sample), and the whole epoch is materialised by ONE gather launch
(``fi_augment2d``, csrc/augment.hip) that is bit-exact with numpy's rot90/flip and scipy's rotate.  Batches are views.

Same names and argument meaning as the reference: ``BaseDataSets(base_dir, split, transform, client, sup_type,
img_class)``, ``RandomGenerator(output_size, img_class)``, ``TwoStreamBatchSampler``; ``DeviceLoader`` stands where
``torch.utils.data.DataLoader(db, batch_size, shuffle, ...)`` stands in flower_pCE_2D.py:303-306 and yields the same
``{'image', 'label', 'idx'}`` dicts (tensors on the device).
"""
from __future__ import annotations

import itertools
import os
import random

import numpy as np
import torch

from .. import _lib as L

ROT_CVAL = {"faz": (0.8, 2), "odoc": (0.0, 3), "polyp": (0.0, 2)}      # dataset.py:213-227 (image, label)
N_DOMAINS = {"faz": 5, "odoc": 5, "polyp": 4}                           # dataset.py:98-175


class _Resident:
    """A decoded split in device memory: images fp32 [n][C][H][W], labels uint8 [n][H][W]."""

    def __init__(self, images, labels, squeeze_channel):
        self.images, self.labels, self.squeeze_channel = images, labels, squeeze_channel

    def __len__(self):
        return self.images.shape[0]


_h5mini_noted = False


def _note_h5mini():
    """Said once per process: the data files are read by the bundled subset decoder, not by h5py."""
    global _h5mini_noted
    if not _h5mini_noted:
        _h5mini_noted = True
        import logging
        logging.getLogger("fedicra_amd").warning(
            "h5py is not installed: reading the HDF5 data files with the bundled read-only subset decoder "
            "(fedicra_amd.dataloaders.h5mini; anything outside the subset raises H5Error)")


class BaseDataSets:
    """dataset.py:63-187.  ``data_list`` holds ``{'image', 'label'}`` numpy pairs exactly like the reference; train
    labels come from ``sup_type`` (e.g. 'scribble'), validation labels from 'mask' (:86-96)."""

    def __init__(self, base_dir=None, split="train", transform=None, client="client1", sup_type="label", img_class="odoc"):
        self._base_dir, self.split, self.img_class = base_dir, split, img_class
        self.sup_type, self.transform = sup_type, transform
        self.sample_list, self.data_list = [], []
        if base_dir is None:
            return
        train_ids, val_ids = self._get_client_ids(client)
        self.sample_list = train_ids if split == "train" else val_ids if split == "val" else []
        print("total {} samples".format(len(self.sample_list)))
        try:
            import h5py as h5
        except ImportError:                                        # this image: the bundled decoder of the files' HDF5
            from . import h5mini as h5                             # subset (raises H5Error on anything outside it)
            _note_h5mini()
        for case in self.sample_list:
            with h5.File(self._base_dir + "/{}".format(case), "r") as h5f:
                image = h5f["image"][:]
                if split == "train":
                    if sup_type == "random_walker":
                        raise NotImplementedError("sup_type='random_walker' (skimage random_walker pseudo labels, "
                                                  "dataset.py:17-60) is not part of this build")
                    label = h5f[sup_type][:]
                else:
                    label = h5f["mask"][:]
            self.data_list.append({"image": image, "label": label})

    @classmethod
    def from_arrays(cls, images, labels, split="train", transform=None, img_class="faz"):
        """In-memory construction (synthetic data, tests): images [n][H][W] (faz) or [n][C][H][W], labels [n][H][W]."""
        self = cls(None, split=split, transform=transform, img_class=img_class)
        self.sample_list = ["mem/{}".format(i) for i in range(len(images))]
        self.data_list = [{"image": np.asarray(im), "label": np.asarray(lb)} for im, lb in zip(images, labels)]
        return self

    def _get_client_ids(self, client):
        """dataset.py:98-175: Domain{d}/train and Domain{d}/test listings; 'client_all' concatenates the domains."""
        nd = N_DOMAINS[self.img_class]
        tr = {d: ["Domain{}/train/{}".format(d, f) for f in os.listdir(self._base_dir + "/Domain{}/train".format(d))]
              for d in range(1, nd + 1)}
        te = {d: ["Domain{}/test/{}".format(d, f) for f in os.listdir(self._base_dir + "/Domain{}/test".format(d))]
              for d in range(1, nd + 1)}
        if client == "client_all":
            return [sum((tr[d] for d in range(1, nd + 1)), []), sum((te[d] for d in range(1, nd + 1)), [])]
        if client.startswith("client") and client[6:].isdigit() and 1 <= int(client[6:]) <= nd:
            return [tr[int(client[6:])], te[int(client[6:])]]
        raise KeyError("ERROR KEY: {}".format(client))               # the reference returns the string "ERROR KEY"

    def __len__(self):
        return len(self.sample_list)

    def __getitem__(self, idx):
        sample = self.data_list[idx]
        if self.split == "train" and self.transform:
            sample = self.transform(sample)
        sample = dict(sample)
        sample["idx"] = idx
        return sample

    def resident(self, device="cuda") -> _Resident:
        """Upload the decoded split once.  All samples must share one shape (they do: 256^2 faz, 384^2 odoc/polyp)."""
        imgs = np.stack([np.asarray(s["image"], dtype=np.float32) for s in self.data_list])
        labs = np.stack([np.asarray(s["label"]).astype(np.uint8) for s in self.data_list])
        squeeze = imgs.ndim == 3
        if squeeze:
            imgs = imgs[:, None]
        return _Resident(torch.from_numpy(imgs).to(device), torch.from_numpy(labs).to(device), squeeze)


class RandomGenerator:
    """dataset.py:231-256.  ``draw`` consumes python ``random`` / ``np.random`` exactly as one reference ``__call__``
    does; ``params`` turns draws into the gather's per-sample tables; ``__call__`` keeps the per-sample protocol."""

    def __init__(self, output_size, img_class="odoc"):
        self.output_size, self.img_class = output_size, img_class

    def draw(self):
        k, axis, angle = -1, 0, None
        if random.random() > 0.5:                                    # :243 -> random_rot_flip :190-207
            k = int(np.random.randint(0, 4))
            axis = int(np.random.randint(0, 2)) if self.img_class == "faz" else int(np.random.randint(1, 3)) - 1
        if random.random() > 0.5:                                    # :245 -> random_rotate :210-228
            angle = int(np.random.randint(-45, 45))
        return k, axis, angle

    @staticmethod
    def params(sources, draws, H, W):
        """-> (ip int32 [n][4], dp fp64 [n][6]) for fi_augment2d.  Matrix / offset as scipy.ndimage.rotate builds them
        (reshape=False): [[c, s], [-s, c]] from cosdg / sindg, offset = centre - M @ centre."""
        from scipy import special
        n = len(draws)
        ip, dp = np.zeros((n, 4), np.int32), np.zeros((n, 6), np.float64)
        for b, (src, (k, axis, angle)) in enumerate(zip(sources, draws)):
            if k >= 0 and (k & 1) and H != W:
                raise ValueError("rot90 by an odd k changes the shape of a {}x{} image".format(H, W))
            ip[b] = (src, k, axis, angle is not None)
            if angle is not None:
                c, s = special.cosdg(angle), special.sindg(angle)
                M = np.array([[c, s], [-s, c]])
                shp = np.asarray([H, W])
                off = (shp - 1) / 2 - M @ ((shp - 1) / 2)
                dp[b] = (M[0, 0], M[0, 1], M[1, 0], M[1, 1], off[0], off[1])
        return ip, dp

    def apply(self, res: _Resident, sources, draws):
        """One launch: sample sources[b] of the resident set, transformed by draws[b] -> (images, labels) on the device."""
        n, C, H, W = res.images.shape
        ip, dp = self.params(sources, draws, H, W)
        dev = res.images.device
        out_img = torch.empty((len(draws), C, H, W), dtype=torch.float32, device=dev)
        out_lab = torch.empty((len(draws), H, W), dtype=torch.uint8, device=dev)
        ci, cl = ROT_CVAL[self.img_class]
        L.augment2d(res.images, res.labels, torch.from_numpy(ip).to(dev), torch.from_numpy(dp).to(dev), out_img, out_lab,
                    ci, cl)
        return (out_img.squeeze(1) if res.squeeze_channel else out_img), out_lab

    def __call__(self, sample):
        ds = BaseDataSets.from_arrays([sample["image"]], [sample["label"]], img_class=self.img_class)
        img, lab = self.apply(ds.resident(), [0], [self.draw()])
        return {"image": img[0], "label": lab[0]}


class DeviceLoader:
    """Stands where ``DataLoader(db, batch_size, shuffle, num_workers=4, pin_memory=True)`` stands
    (flower_pCE_2D.py:303-306).  Per epoch: the permutation is drawn the way torch's RandomSampler draws it (a seed from
    the default torch generator, then ``randperm``), the augmentation draws follow in batch order (the order a
    ``num_workers=0`` DataLoader makes them), and one launch materialises the epoch; batches are views of it."""

    def __init__(self, dataset: BaseDataSets, batch_size=1, shuffle=False, device="cuda", drop_last=False):
        self.dataset, self.batch_size, self.shuffle, self.drop_last = dataset, int(batch_size), shuffle, drop_last
        self.res = dataset.resident(device)
        tf = dataset.transform if dataset.split == "train" else None
        for t in getattr(tf, "transforms", [tf] if tf is not None else []):      # transforms.Compose([RandomGenerator])
            if not isinstance(t, RandomGenerator):
                raise TypeError("DeviceLoader only understands RandomGenerator transforms")
            tf = t
        self.transform = tf

    def __len__(self):
        n = len(self.res)
        return n // self.batch_size if self.drop_last else -(-n // self.batch_size)

    def __iter__(self):
        n = len(self.res)
        if self.shuffle:
            torch.empty((), dtype=torch.int64).random_()          # the DataLoader iterator's base seed (workers'), drawn first
            g = torch.Generator()
            g.manual_seed(int(torch.empty((), dtype=torch.int64).random_().item()))     # RandomSampler.__iter__
            order = torch.randperm(n, generator=g).tolist()
        else:
            torch.empty((), dtype=torch.int64).random_()
            order = list(range(n))
        if self.transform is not None:
            draws = [self.transform.draw() for _ in order]
            images, labels = self.transform.apply(self.res, order, draws)
        else:
            sel = torch.tensor(order, device=self.res.images.device)
            images, labels = self.res.images[sel], self.res.labels[sel]
            if self.res.squeeze_channel:
                images = images.squeeze(1)
        idx = torch.tensor(order)
        for b in range(len(self)):
            s = slice(b * self.batch_size, min(n, (b + 1) * self.batch_size))
            yield {"image": images[s], "label": labels[s], "idx": idx[s]}


class TwoStreamBatchSampler:
    """dataset.py:259-289: batches of (primary, secondary) indices; one epoch = one pass over the primary indices."""

    def __init__(self, primary_indices, secondary_indices, batch_size, secondary_batch_size):
        self.primary_indices, self.secondary_indices = primary_indices, secondary_indices
        self.secondary_batch_size = secondary_batch_size
        self.primary_batch_size = batch_size - secondary_batch_size
        assert len(self.primary_indices) >= self.primary_batch_size > 0
        assert len(self.secondary_indices) >= self.secondary_batch_size > 0

    def __iter__(self):
        primary_iter = iterate_once(self.primary_indices)
        secondary_iter = iterate_eternally(self.secondary_indices)
        return (p + s for p, s in zip(grouper(primary_iter, self.primary_batch_size),
                                      grouper(secondary_iter, self.secondary_batch_size)))

    def __len__(self):
        return len(self.primary_indices) // self.primary_batch_size


def iterate_once(iterable):
    return np.random.permutation(iterable)


def iterate_eternally(indices):
    def infinite_shuffles():
        while True:
            yield np.random.permutation(indices)
    return itertools.chain.from_iterable(infinite_shuffles())


def grouper(iterable, n):
    args = [iter(iterable)] * n
    return zip(*args)
