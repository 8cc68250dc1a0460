"""Fixtures for fedicra_amd/dataloaders/h5mini.py (run in the build container only; test infrastructure).

    python -m oracle.gen_h5_fixtures

The reference reads its data sets with h5py (/root/reference/code/dataloaders/dataset.py:84-96), which this image does not
have, so the expected arrays are decoded INDEPENDENTLY of any HDF5 code: the reference ships the PNG originals next to the
.h5 files (data/FAZ/DomainN/{train,test}/{imgs,mask,scr,scr_n,keypoint,block,box}, data/ODOC/...), and over a sweep of
the data directory the relation is exact:

    image   = float32 min-max normalisation of the PNG  ((p - p.min()) / (p.max() - p.min()); channel 0 of the grey FAZ
              PNGs as [H][W], the RGB ODOC PNGs as [3][H][W])
    labels  = the PNG grey level through a fixed table (FAZ 0/127/254 -> 0/1/2, ODOC 0/170/85/255 -> 0/1/2/3)

Written: the four smallest .h5 data files of the reference (one per data set and split; data files, copied byte for byte
into tests/golden/h5/) and tests/golden/g21_h5_decoded.npz with the arrays built from the PNGs (images kept as the uint8
PNG planes; the test applies the normalisation).  tests/test_h5mini_cpu.py decodes the .h5 files with h5mini and must
reproduce those arrays bit for bit; where /root/reference is present the same test sweeps a sample of all 3115 files.
"""
from __future__ import annotations

import glob
import os
import shutil

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")
DATA = "/root/reference/data"
PNG_DIR = {"image": "imgs", "mask": "mask", "scribble": "scr", "scribble_noisy": "scr_n", "keypoint": "keypoint",
           "block": "block", "box": "box"}
LABEL_TABLE = {"FAZ": {0: 0, 127: 1, 254: 2}, "ODOC": {0: 0, 170: 1, 85: 2, 255: 3}}


def png_arrays(dataset, domain, split, stem):
    """name -> array as the .h5 file must decode ('image' stays the uint8 PNG planes; see ``normalise``)."""
    from PIL import Image
    out = {}
    for name, sub in PNG_DIR.items():
        path = "{}/{}/{}/{}/{}/{}.png".format(DATA, dataset, domain, split, sub, stem)
        if not os.path.exists(path):
            continue
        p = np.asarray(Image.open(path))
        if name == "image":
            out[name] = p[..., 0].copy() if dataset == "FAZ" else p[..., :3].transpose(2, 0, 1).copy()
        else:
            g = p[..., 0] if p.ndim == 3 else p
            lut = np.full(256, 255, np.uint8)
            for k, v in LABEL_TABLE[dataset].items():
                lut[k] = v
            out[name] = lut[g]
    return out


def normalise(planes):
    p = planes.astype(np.float32)
    return (p - p.min()) / (p.max() - p.min())


def smallest(dataset, split):
    files = glob.glob("{}/{}_h5/Domain*/{}/*.h5".format(DATA, dataset, split))
    return min(files, key=lambda f: (os.path.getsize(f), f))


def main():
    os.makedirs(os.path.join(OUT, "h5"), exist_ok=True)
    arrays, names = {}, []
    for dataset in ("FAZ", "ODOC"):
        for split in ("train", "test"):
            src = smallest(dataset, split)
            domain, stem = src.split("/")[-3], os.path.basename(src)[:-3]
            tag = "{}_{}_{}".format(dataset.lower(), split, stem)
            shutil.copyfile(src, os.path.join(OUT, "h5", tag + ".h5"))
            os.chmod(os.path.join(OUT, "h5", tag + ".h5"), 0o644)
            names.append(tag)
            for k, a in png_arrays(dataset, domain, split, stem).items():
                arrays["{}/{}".format(tag, k)] = a
            print(tag, os.path.getsize(src), "bytes", sorted(k for k in arrays if k.startswith(tag)))
    np.savez_compressed(os.path.join(OUT, "g21_h5_decoded.npz"), files=np.array(names), **arrays)


if __name__ == "__main__":
    main()
