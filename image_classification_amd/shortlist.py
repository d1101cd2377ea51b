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
    """Rows and labels of a fitted index as one .npz (metric and weights by name)."""
    metric = index.metric.__name__ if callable(index.metric) else str(index.metric)
    np.savez(path, fit_x=np.asarray(index._fit_X), labels=np.asarray(index._labels),
             neighbors=np.int64(index.n_neighbors), metric=np.str_(metric),
             weights=np.str_(index.weights))


def load_shortlist_index(path, backend=None):
    from . import clustering
    data = np.load(path if str(path).endswith(".npz") else str(path) + ".npz", allow_pickle=False)
    metric = str(data["metric"])
    metric = getattr(clustering, metric) if metric in ("intersection", "chisqr") else metric
    index = KNNScikit(neighbors=int(data["neighbors"]), metric=metric,
                      weights=str(data["weights"]), backend=backend)
    index.train(data["fit_x"], list(data["labels"]))
    return index
