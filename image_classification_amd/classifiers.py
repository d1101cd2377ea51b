"""`KNNScikit` with the reference's interface, brute-force Euclidean search on the device.

Mirrors model/classifiers.py:165-197: `KNNScikit(neighbors, metric, weights, **kw)`,
`train(features, labels)` fits once (later calls are ignored, the `_is_trained` guard at :175),
`get_neighbors(features, neighbors, dist=True)` returns what
`KNeighborsClassifier.kneighbors(features, n_neighbors, return_distance)` returns: distances
float64 [M, k] ascending and indices int64 [M, k], or the indices alone.

Only what the descriptor-matching path uses is implemented: metric 'euclidean' (or 'minkowski'
with p=2) and at most two neighbours (model/robust_matching.py:73-80).  Anything else raises
NotImplementedError rather than running on the CPU.
"""
import numpy as np

from .backend import default_backend

__all__ = ["KNNScikit"]


class KNNScikit(object):

    def __init__(self, neighbors=3, metric="euclidean", weights="distance", backend=None,
                 **distance_kwargs):
        self._is_trained = False
        self.n_neighbors = neighbors
        self.metric = metric
        self.weights = weights
        self.distance_kwargs = distance_kwargs
        self._backend = backend
        self._fit_X = None
        self._cb = None
        p = distance_kwargs.get("p", 2)
        if not (metric == "euclidean" or (metric == "minkowski" and p == 2)):
            raise NotImplementedError("KNNScikit on the device supports the euclidean metric only "
                                      "(got %r)" % (metric,))

    def _b(self):
        return self._backend or default_backend()

    def train(self, features, labels):
        if not self._is_trained:
            features = np.asarray(features)
            if features.ndim != 2:
                raise ValueError("Expected 2D array, got %dD array instead" % features.ndim)
            labels = list(labels)
            if len(labels) != features.shape[0]:
                raise ValueError("Found input variables with inconsistent numbers of samples: "
                                 "[%d, %d]" % (features.shape[0], len(labels)))
            self.labels_map = sorted(list(set(labels)))
            self._labels = np.asarray(labels)
            self._fit_X = features
            self._cb = self._b().codebook(features)
            self._is_trained = True

    def get_neighbors(self, features, neighbors, dist=True):
        if not self._is_trained:
            raise RuntimeError("This KNNScikit instance is not fitted yet.")
        neighbors = int(neighbors)
        n_fit = self._fit_X.shape[0]
        if neighbors < 1:
            raise ValueError("Expected n_neighbors > 0. Got %d" % neighbors)
        if neighbors > n_fit:
            raise ValueError("Expected n_neighbors <= n_samples_fit, but n_neighbors = %d, "
                             "n_samples_fit = %d, n_samples = %d"
                             % (neighbors, n_fit, np.asarray(features).shape[0]))
        if neighbors > 2:
            raise NotImplementedError("the device search returns at most 2 neighbours")
        features = np.asarray(features)
        if features.ndim != 2 or features.shape[1] != self._fit_X.shape[1]:
            raise ValueError("X has %s features, but KNNScikit is expecting %d features as input."
                             % (features.shape[1:] or "?", self._fit_X.shape[1]))
        b = self._b()
        if features.shape[0] == 0:
            ind = np.empty((0, neighbors), dtype=np.int64)
            return (np.empty((0, neighbors), dtype=np.float64), ind) if dist else ind
        i1, d1, i2, d2, _ = b.assign_top2_np(features, self._cb)
        if neighbors == 1:
            ind = i1.cpu().numpy().astype(np.int64)[:, None]
            dd = d1.cpu().numpy().astype(np.float64)[:, None]
        else:
            ind = np.stack([i1.cpu().numpy(), i2.cpu().numpy()], axis=1).astype(np.int64)
            dd = np.stack([d1.cpu().numpy(), d2.cpu().numpy()], axis=1).astype(np.float64)
        return (dd, ind) if dist else ind

    def test(self, features):
        raise NotImplementedError("label prediction is outside the descriptor-matching path")
