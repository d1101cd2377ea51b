"""`compute_vocab_features` with the reference's signature.

Mirrors vision/utils/scene_functions.py:375-392: `codes[i] = argmin_j |feat_i - vocab_j|^2` as an
int64 array (the reference allocates `np.zeros(k, dtype=np.int)`), first minimum on ties.  The
reference walks the rows in batches of `vocab.shape[0]` through sklearn's euclidean_distances to
bound host memory; the device kernel never materialises the distance matrix, so one call covers
all rows.
"""
import numpy as np

from .backend import default_backend

__all__ = ["compute_vocab_features"]


def compute_vocab_features(vocab, feat, backend=None):
    vocab = np.asarray(vocab)
    feat = np.asarray(feat)
    if vocab.ndim != 2 or feat.ndim != 2:
        raise ValueError("vocab and feat must be 2-D")
    if vocab.shape[1] != feat.shape[1]:
        raise ValueError("Incompatible dimension for X and Y matrices: X.shape[1] == %d while "
                         "Y.shape[1] == %d" % (feat.shape[1], vocab.shape[1]))
    if feat.shape[0] == 0:
        return np.zeros(0, dtype=np.int64)
    b = backend or default_backend()
    idx, _ = b.assign_top1(b.f32(feat), b.codebook(vocab), want_dist=False)
    return idx.cpu().numpy().astype(np.int64)
