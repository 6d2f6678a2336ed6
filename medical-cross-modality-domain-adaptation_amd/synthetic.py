"""Synthetic 256x256x3 slices in the reference's tfrecord layout: z-scored-looking images, blob label maps with every
class present.  Seeds: 0 = "MR", 1 = "CT"."""
import os

import numpy as np

from .tfrecord import write_slice


def blob_labels(rng, n_class=5, size=256):
    yy, xx = np.mgrid[0:size, 0:size]
    lab = np.zeros((size, size), np.float32)
    for c in range(1, n_class):
        cy, cx = rng.integers(size // 6, size - size // 6, 2)
        ry, rx = rng.integers(max(size // 20, 1), max(size // 6, 2), 2)
        lab[((yy - cy) / ry) ** 2 + ((xx - cx) / rx) ** 2 <= 1] = c
    return lab


def make_slice(rng, n_class=5, size=256):
    lab = blob_labels(rng, n_class, size)
    img = rng.standard_normal((size, size, 3)).astype(np.float32) + 0.3 * lab[:, :, None]
    label_vol = np.repeat(lab[:, :, None], 3, axis=2).astype(np.float32)     # label_vol holds 3 slices; the middle one is used
    return img.astype(np.float32), label_vol


def write_dataset(folder, n, seed=0, prefix="slice", size=256):
    """writes n one-record .tfrecords files and a list file; returns the list of paths"""
    os.makedirs(folder, exist_ok=True)
    rng = np.random.default_rng(seed)
    paths = []
    for i in range(n):
        img, lab = make_slice(rng, size=size)
        p = os.path.join(folder, "%s_%04d.tfrecords" % (prefix, i))
        write_slice(p, img, lab)
        paths.append(p)
    with open(os.path.join(folder, prefix + "_list"), "w") as f:
        f.write("\n".join(paths) + "\n")
    return paths
