"""Histogram-space shortlist index of the logo pipeline.

The reference builds it in `KnnTrainer.train` (vision/logo_recognition/knn_training.py:34-60):
bag-of-words histograms of the training logos become the rows of a `KNNScikit`.  That class is
orchestration only -- it calls `bow.extract` and `KNNScikit.train` -- so it keeps working
unchanged on top of this package once its two imports point here; what this
module offers instead is the same step as two functions plus a pickle-free on-disk form.
"""
import numpy as np

from .classifiers import KNNScikit
from .clustering import intersection

__all__ = ["fit_shortlist_index", "save_shortlist_index", "load_shortlist_index"]


def fit_shortlist_index(bow, paths, labels, neighbors=3, metric=intersection, weights="distance",
                        backend=None):
    """Quantise `paths` with `bow` (one device call for all of them) and index the histograms.

    Returns (index, image_descs, hists): `index.get_neighbors(hist, k, dist=False)` is the
    shortlist step of vision/logo_recognition/matching_flow.py:23."""
    labels = list(labels)
    if len(labels) != len(paths):
        raise ValueError("%d paths but %d labels" % (len(paths), len(labels)))
    image_descs, hists = bow.extract(list(paths))
    index = KNNScikit(neighbors=neighbors, metric=metric, weights=weights, backend=backend)
    index.train(hists, labels)
    return index, image_descs, hists


def save_shortlist_index(index, path):
    """Rows, labels and constructor arguments of a fitted index as one numpy archive
    (`KNNScikit.save_to_file`; the archive is written under exactly `path`)."""
    index.save_to_file(path)


def load_shortlist_index(path, backend=None):
    import os
    path = str(path)
    if not os.path.exists(path) and os.path.exists(path + ".npz"):
        path += ".npz"           # archives written by numpy.savez under a name without suffix
    return KNNScikit.load_from_file(path, backend=backend)
