"""`KnnTrainer` with the reference's interface (vision/logo_recognition/knn_training.py:8-60).

Builds the histogram-space shortlist index of the logo pipeline: the bag-of-words histograms of
the training logos (`bow.extract`, one device call for all logos) become the rows of a
`KNNScikit` whose searches run on the device.  Model persistence (:62-90 pickles the scikit-learn
object) is not part of the search path; `save_knn` / `load_knn` keep the rows and labels in a
plain .npz instead.
"""
import os

import numpy as np

from .classifiers import KNNScikit
from .clustering import intersection

__all__ = ["KnnTrainer"]


class KnnTrainer(object):

    def __init__(self, bag_of_words, metric=intersection, weights="distance", neighbors=3,
                 algorithm="auto", backend=None):
        self.metric = metric
        self.neighbors = neighbors
        self.algorithm = algorithm   # accepted for compatibility; the device search is brute force
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
                                 weights=self.weights, backend=self._backend)
        logo_target_label_ids = [l.target_label_id for l in self.logos]
        self._knnsck.train(self.logo_hists, logo_target_label_ids)

    def save_knn(self, model_dir):
        np.savez(self._knn_path(model_dir) + ".npz", fit_x=np.asarray(self.logo_hists),
                 labels=np.asarray([l.target_label_id for l in self.logos]))

    def load_knn(self, model_dir):
        data = np.load(self._knn_path(model_dir) + ".npz", allow_pickle=False)
        self._knnsck = KNNScikit(neighbors=self.neighbors, metric=self.metric,
                                 weights=self.weights, backend=self._backend)
        self._knnsck.train(data["fit_x"], list(data["labels"]))
        return self._knnsck

    @staticmethod
    def _knn_path(model_dir):
        return os.path.join(model_dir, "knn")

    @staticmethod
    def _logos_path(model_dir):
        return os.path.join(model_dir, "logos")
