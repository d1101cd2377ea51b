"""`KNNScikit` with the reference's interface, brute-force search on the device.

Mirrors model/classifiers.py:165-197: `KNNScikit(neighbors, metric, weights, **kw)`,
`train(features, labels)` fits once (later calls are ignored, the `_is_trained` guard at :175),
`get_neighbors(features, neighbors, dist=True)` returns what
`KNeighborsClassifier.kneighbors(features, n_neighbors, return_distance)` returns: distances
float64 [M, k] ascending and indices int64 [M, k], or the indices alone.  `test` and
`test_probability` (:180-193) vote over the neighbour lists as scikit-learn does.

Two searches sit behind it:
  * Euclidean, width <= 128, at most two neighbours (the descriptor matching of
    model/robust_matching.py:73-80): the matrix-core distance / top-2 kernel;
  * everything else (bag-of-words histograms of any width, 'manhattan' as configured in
    vision/logo_recognition/training.py:41, the `intersection` / `chisqr` functions of
    model/clustering.py passed as `metric`, more neighbours): distance matrix + row top-k.
Metrics without a device kernel raise NotImplementedError rather than running on the CPU.
"""
import numpy as np

from . import clustering as _clustering
from .backend import default_backend

__all__ = ["KNNScikit"]

_STRING_METRICS = {"euclidean": "euclidean", "l2": "euclidean", "manhattan": "manhattan",
                   "cityblock": "manhattan", "l1": "manhattan", "sqeuclidean": "sqeuclidean"}


def _device_metric(metric, kwargs):
    if callable(metric):
        if metric is _clustering.intersection:
            return "intersection"
        if metric is _clustering.chisqr:
            return "chisqr"
        raise NotImplementedError("only the `intersection` and `chisqr` functions of this package "
                                  "are supported as callable metrics")
    if metric == "minkowski":
        p = kwargs.get("p", 2)
        if p == 2:
            return "euclidean"
        if p == 1:
            return "manhattan"
        raise NotImplementedError("minkowski is supported for p = 1 and p = 2 only")
    if metric in _STRING_METRICS:
        return _STRING_METRICS[metric]
    raise NotImplementedError("KNNScikit on the device supports euclidean, manhattan, sqeuclidean, "
                              "intersection and chisqr (got %r)" % (metric,))


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
        self._metric = _device_metric(metric, distance_kwargs)
        if weights not in ("distance", "uniform"):
            raise NotImplementedError("weights must be 'distance' or 'uniform'")

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
            self.classes_ = np.asarray(self.labels_map)
            self._fit_X = features
            self._cb = self._b().codebook(features)
            self._is_trained = True

    # Evaluation helpers of the reference (model/classifiers.py:199-215) over `DataBag`-like
    # objects with `.feats` and `.labs`; accuracy is the fraction of equal labels, as
    # sklearn.metrics.accuracy_score and KNeighborsClassifier.score compute it.
    @classmethod
    def score_proba(cls, train_bag, test_bag, neg_bag, min_prob_diff=0.4, **kwargs):
        knn = cls(**kwargs)
        knn.train(train_bag.feats, train_bag.labs)
        y_test_pred = knn.test_probability(test_bag.feats, min_prob_diff)
        test_score = float(np.mean(np.asarray(test_bag.labs) == np.asarray(y_test_pred)))
        y_neg_pred = knn.test_probability(neg_bag.feats, min_prob_diff)
        neg_score = 1 - float(np.mean(np.asarray(neg_bag.labs) == np.asarray(y_neg_pred)))
        return test_score, neg_score

    @classmethod
    def score(cls, train_bag, test_bag, **kwargs):
        knn = cls(**kwargs)
        knn.train(train_bag.feats, train_bag.labs)
        return float(np.mean(knn.test(test_bag.feats) == np.asarray(test_bag.labs)))

    def train_and_test(self, features, labels):
        """model/classifiers.py:49-51"""
        self.train(features, labels)
        return self.test(features)

    # The reference keeps a fitted index as a pickle of the object (model/classifiers.py:53-63;
    # vision/logo_recognition/model.py:23 loads the model directory's `knn` file with it).  Same
    # method names here, but the file is a numpy archive of the fitted rows, the labels and the
    # constructor arguments -- written under exactly the name given -- and nothing is unpickled.
    def save_to_file(self, file_name):
        if not self._is_trained:
            raise RuntimeError("This KNNScikit instance is not fitted yet.")
        import json
        metric = self.metric.__name__ if callable(self.metric) else str(self.metric)
        kwargs = {k: v for k, v in self.distance_kwargs.items()
                  if isinstance(v, (int, float, str, bool)) or v is None}
        with open(file_name, "wb") as f:
            np.savez(f, fit_x=np.asarray(self._fit_X), labels=np.asarray(self._labels),
                     neighbors=np.int64(self.n_neighbors), metric=np.str_(metric),
                     weights=np.str_(self.weights), kwargs=np.str_(json.dumps(kwargs)))

    @classmethod
    def load_from_file(cls, file_name, backend=None):
        import json
        with open(file_name, "rb") as f:
            data = np.load(f, allow_pickle=False)
            metric = str(data["metric"])
            if metric in ("intersection", "chisqr"):
                metric = getattr(_clustering, metric)
            kwargs = json.loads(str(data["kwargs"])) if "kwargs" in data.files else {}
            index = cls(neighbors=int(data["neighbors"]), metric=metric,
                        weights=str(data["weights"]), backend=backend, **kwargs)
            index.train(data["fit_x"], list(data["labels"]))
        return index

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
        features = np.asarray(features)
        if features.ndim == 1 and features.shape[0] == self._fit_X.shape[1]:
            # one sample given as a vector: the logo flow hands the histogram of its one image to
            # this step that way (vision/logo_recognition/matching_flow.py:37-38, then
            # `x[0].tolist()` on the result), which the scikit-learn of the reference's time took
            # as a single row; later versions refuse it
            features = features[None, :]
        if features.ndim != 2 or features.shape[1] != self._fit_X.shape[1]:
            raise ValueError("X has %s features, but KNNScikit is expecting %d features as input."
                             % (features.shape[1:] or "?", self._fit_X.shape[1]))
        b = self._b()
        if features.shape[0] == 0:
            ind = np.empty((0, neighbors), dtype=np.int64)
            return (np.empty((0, neighbors), dtype=np.float64), ind) if dist else ind
        if self._metric == "euclidean" and neighbors <= 2 and features.shape[1] <= 128:
            i1, d1, i2, d2, _ = b.assign_top2_np(features, self._cb)
            if neighbors == 1:
                ind = i1.cpu().numpy().astype(np.int64)[:, None]
                dd = d1.cpu().numpy().astype(np.float64)[:, None]
            else:
                ind = np.stack([i1.cpu().numpy(), i2.cpu().numpy()], axis=1).astype(np.int64)
                dd = np.stack([d1.cpu().numpy(), d2.cpu().numpy()], axis=1).astype(np.float64)
        else:
            dd, ind = b.knn_rows(b.f32(features), self._cb.c, neighbors, self._metric)
            dd = dd.cpu().numpy()
            ind = ind.cpu().numpy().astype(np.int64)
        return (dd, ind) if dist else ind

    # -- label votes over the neighbour lists (host side: [M, k] arrays) ---------------------------
    def _votes(self, features):
        dd, ind = self.get_neighbors(features, self.n_neighbors, dist=True)
        w = np.ones_like(dd)
        if self.weights == "distance":
            with np.errstate(divide="ignore"):
                w = 1.0 / dd
            zero = np.isinf(w)
            rows = zero.any(axis=1)
            w[rows] = zero[rows].astype(np.float64)
        col = np.searchsorted(self.classes_, self._labels[ind])
        votes = np.zeros((dd.shape[0], len(self.classes_)), dtype=np.float64)
        rows = np.arange(dd.shape[0])
        for j in range(ind.shape[1]):
            np.add.at(votes, (rows, col[:, j]), w[:, j])
        return votes

    def predict_proba(self, features):
        votes = self._votes(features)
        s = votes.sum(axis=1, keepdims=True)
        s[s == 0] = 1.0
        return votes / s

    def test(self, features):
        """Predicted label per row: the class with the largest vote, lowest class on ties."""
        return self.classes_[np.argmax(self._votes(features), axis=1)]

    def test_probability(self, features, min_prob_diff):
        """Best label when its probability leads the runner-up by `min_prob_diff`, else -1
        (model/classifiers.py:183-193)."""
        y_pred = []
        for row in self.predict_proba(features):
            idx = row.argsort()
            if row[idx[-1]] - row[idx[-2]] >= min_prob_diff:
                y_pred.append(self.labels_map[idx[-1]])
            else:
                y_pred.append(-1)
        return y_pred
