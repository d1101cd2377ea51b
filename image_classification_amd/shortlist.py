"""Histogram-space shortlist index of the logo pipeline.

The reference builds it in `KnnTrainer.train` (vision/logo_recognition/knn_training.py:34-60):
bag-of-words histograms of the training logos become the rows of a `KNNScikit`.  That class is
orchestration only -- it calls `bow.extract` and `KNNScikit.train`.  It is kept here under its own
name with its own methods, beside the same step as two functions and a pickle-free on-disk form.
"""
import numpy as np

from .classifiers import KNNScikit
from .clustering import intersection

__all__ = ["KnnTrainer", "fit_shortlist_index", "save_shortlist_index", "load_shortlist_index"]


class KnnTrainer(object):
    """The reference's trainer under its own name and methods
    (vision/logo_recognition/knn_training.py:8-90): `train(logos)` quantises the logos' images
    with the bag of words in one device call, hangs histogram and descriptors on every logo
    object (`feature`, `image_desc`) and fits a `KNNScikit` on the histograms with the logos'
    `target_label_id`s; `save_knn(model_dir)` writes the model directory's `knn` file.
    `save_logos` writes the logos' arrays as a numpy archive instead of pickling the objects."""

    def __init__(self, bag_of_words, metric=intersection, weights="distance", neighbors=3,
                 algorithm="auto", backend=None):
        self.metric = metric
        self.neighbors = neighbors
        self.algorithm = algorithm
        self.weights = weights
        self.bow = bag_of_words
        self._backend = backend

    def _get_logo_hists(self):
        logo_paths = [l.path for l in self.logos]
        image_desc, self.logo_hists = self.bow.extract(logo_paths)
        for logo, h, img_d in zip(self.logos, self.logo_hists, image_desc):
            logo.feature = h
            logo.image_desc = img_d

    def train(self, logos):
        self.logos = logos
        self._get_logo_hists()
        self._knnsck = KNNScikit(neighbors=self.neighbors, metric=self.metric,
                                 weights=self.weights, algorithm=self.algorithm,
                                 backend=self._backend)
        self._knnsck.train(self.logo_hists, [l.target_label_id for l in self.logos])

    def save_knn(self, model_dir):
        self._knnsck.save_to_file(self._knn_path(model_dir))

    def save_logos(self, model_dir):
        """`logos` of the model directory: paths, label ids, histograms and every logo's
        descriptors / keypoints (concatenated, with row offsets) as arrays -- no pickle."""
        descs = [l.image_desc.descriptors for l in self.logos]
        rows = np.cumsum([0] + [0 if d is None else len(d) for d in descs])
        some = [d for d in descs if d is not None and len(d)]
        kps = [np.asarray(l.image_desc.keypoints, dtype=np.float64).reshape(-1, 2)
               for l, d in zip(self.logos, descs) if d is not None and len(d)]
        with open(self._logos_path(model_dir), "wb") as f:
            np.savez(f, paths=np.asarray([str(l.path) for l in self.logos]),
                     target_label_ids=np.asarray([l.target_label_id for l in self.logos]),
                     features=np.asarray(self.logo_hists), row_offsets=rows,
                     descriptors=np.concatenate(some) if some else np.empty((0, 0), np.float32),
                     keypoints=np.concatenate(kps) if kps else np.empty((0, 2)))

    @staticmethod
    def _knn_path(model_dir):
        import os
        return os.path.join(model_dir, "knn")

    @staticmethod
    def _logos_path(model_dir):
        import os
        return os.path.join(model_dir, "logos")


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
