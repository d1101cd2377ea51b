"""The k-means hashing pieces of the face index that sit on the descriptor-search path.

Mirrors vision/facerec/face_indexing.py:131-170 and the distance re-rank at :468-482:

  KMeansHashing(codebook, distrib).get_hash(feature)   nearest codeword of one signature (:139-143)
  _quantize(features, num_clusters, iterations)        kmeans2(minit='points') + label histogram
                                                       (:156-161)
  get_n_codebooks(features, n, num_clusters, iters)    n independent codebooks (:166-170; the
                                                       reference maps them over a process pool,
                                                       here they run back to back on the device)
  neighbor_distances / sorted_neighbors                cdist(feature, feats) and the sort of
                                                       FaceIndex.get_neighbors (:470-482)

The LSH tables, the database access and the signature extraction of that module are not part of
this path.
"""
import os

import numpy as np

from .backend import default_backend
from .vq import kmeans2, vq

__all__ = ["KMeansHashing", "get_n_codebooks", "neighbor_distances", "sorted_neighbors"]


def _label_histogram(labels, numbins, backend=None):
    """scipy.stats.histogram(labels, numbins=numbins)[0] as called at :160 (that function was
    removed from scipy 1.0; its default limits put `numbins` equal bins over
    [min - s, max + s] with s = (max - min) / (2 (numbins - 1))).  The label counts come from the
    device histogram kernel; placing at most `numbins` distinct integer labels into those bins is
    host bookkeeping."""
    b = backend or default_backend()
    labels = np.asarray(labels)
    nlab = int(labels.max()) + 1
    counts = b.hist(b.i32(labels), b.i64([0, labels.shape[0]]), nlab, "none").cpu().numpy()[0]
    lo, hi = float(labels.min()), float(labels.max())
    s = (hi - lo) / (2.0 * (numbins - 1.0)) if numbins > 1 else 0.5
    lo, hi = lo - s, hi + s
    binsize = (hi - lo) / float(numbins)
    out = np.zeros(numbins, dtype=np.float64)
    present = np.nonzero(counts)[0]
    if binsize > 0:
        bins = np.floor((present - lo) / binsize).astype(np.int64)
        keep = (bins >= 0) & (bins < numbins)
        np.add.at(out, bins[keep], counts[present][keep])
    else:
        out[0] = counts.sum()
    return out


def _quantize(features, num_clusters, iterations, backend=None, rng=None):
    cb, neigh = kmeans2(features, num_clusters, iter=iterations, minit="points", backend=backend,
                        rng=rng)
    return cb, _label_histogram(neigh, num_clusters, backend)


def quantize(args):
    return _quantize(*args)


def get_n_codebooks(features, num_codebooks, num_clusters, iterations, backend=None, rng=None):
    return [_quantize(features, num_clusters, iterations, backend, rng)
            for _ in range(num_codebooks)]


class KMeansHashing(object):
    def __init__(self, codebook, distrib, backend=None):
        self.hash_size = 1
        self.feature_size = codebook.shape[1]
        self.codebook = codebook
        self.distrib = distrib
        self._backend = backend

    # A face is hashed against every table, one call each (vision/facerec/face_indexing.py:
    # 139-143 through FaceIndex): the codebook is therefore held as a private read-only copy whose
    # device copy is made once, instead of being looked up by a hash of its 4.6 MB (600 x 1937
    # float64) on every call.  Assigning `h.codebook = other` replaces it.
    @property
    def codebook(self):
        return self._codebook.view()

    @codebook.setter
    def codebook(self, value):
        held = np.array(value, copy=True)
        held.setflags(write=False)
        self._codebook = held
        self._resident = None

    def __getstate__(self):
        state = dict(self.__dict__)
        state["_resident"] = None          # device memory does not travel
        return state

    def _cb(self):
        b = self._backend or default_backend()
        key = (os.getpid(), id(b))
        got = getattr(self, "_resident", None)
        if got is None or got[0] != key:
            got = self._resident = (key, b.codebook(self._codebook))
        return got[1]

    def get_hash(self, feature):
        feature = np.reshape(feature, (1, self._codebook.shape[1]))
        code, _dist = vq(feature, self._codebook, backend=self._backend, resident=self._cb())
        return str(code[0])

    def get_hashes(self, features):
        """Hashes of many signatures in one device call (the reference hashes them one by one)."""
        code, _dist = vq(np.asarray(features), self._codebook, backend=self._backend,
                         resident=self._cb())
        return [str(c) for c in code]

    @classmethod
    def create_hash_tables(cls, num_tables, features, num_clusters, iterations=5, backend=None,
                           rng=None):
        return [cls(cb, db, backend) for cb, db in
                get_n_codebooks(features, num_tables, num_clusters, iterations, backend, rng)]


def neighbor_distances(feature, feats, backend=None):
    """scipy.spatial.distance.cdist(feature, feats) (Euclidean), float64 [1 or m, n]."""
    b = backend or default_backend()
    feature = np.atleast_2d(np.asarray(feature))
    feats = np.atleast_2d(np.asarray(feats))
    if feature.shape[1] != feats.shape[1]:
        raise ValueError("XA and XB must have the same number of columns "
                         "(i.e. feature dimension.)")
    return b.pairwise(b.f32(feature), b.f32(feats), "euclidean").cpu().numpy()


def sorted_neighbors(feature, box_ids, label_ids, feats, num_neighbors="all", return_sorted=True,
                     backend=None):
    """The (box_id, label_id, distance) list of FaceIndex.get_neighbors (:470-482)."""
    dists = neighbor_distances(feature, feats, backend)
    neighbors = [(bid, lid, d) for bid, lid, d in zip(box_ids, label_ids, dists[0])]
    if return_sorted:
        neighbors = sorted(neighbors, key=lambda x: x[2])
        assert num_neighbors == "all" or isinstance(num_neighbors, int), \
            'num_neighbors should be either "all" or must be an integer'
        if num_neighbors != "all":
            neighbors = neighbors[:num_neighbors]
    return neighbors
