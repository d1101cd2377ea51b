"""Drop-in for the two scipy.cluster.vq functions the reference calls.

`vq(obs, code_book)`   replaces scipy.cluster.vq.vq at model/features.py:1166 and
                       vision/facerec/face_indexing.py:141
`kmeans2(data, k, ...)` replaces scipy.cluster.vq.kmeans2 at vision/facerec/face_indexing.py:158

Same argument meaning, return types and error behaviour; the arithmetic runs in float32 on the
GPU (scipy promotes to the common dtype of its arguments, float64 for the reference's float64
vocabulary).  The index returned for a row is the one a direct float32 evaluation of
sum (x - c)^2 over the whole code book would pick (lowest index on ties): the matrix-core search
knows the rounding error of the values it ranks by and re-searches, by direct differences, every
row it cannot decide with certainty.  Against scipy's float64 that leaves only the rounding of
the direct sum itself: distances agree to about 1e-7 relative, and indices wherever two code
words are not closer to each other than that.
"""
import numpy as np

from .backend import default_backend
from .clustering import MAX_DIM, kmeans_fit


def _as_2d(a, name):
    a = np.asarray(a)
    if a.ndim == 1:
        a = a[:, None]
    if a.ndim != 2:
        raise ValueError("%s must be 1-D or 2-D" % name)
    return a


def vq(obs, code_book, check_finite=True, backend=None, resident=None):
    """(code int32 [M], dist [M]) -- index of and Euclidean distance to the nearest code.

    `resident`: the backend's device-resident form of `code_book` (`backend.codebook(code_book)`),
    for callers that hold a code book nobody else can edit and search it call after call: the
    lookup of the device copy -- a hash over every byte of `code_book` -- is then skipped."""
    obs_a = np.asarray(obs)
    cb_a = np.asarray(code_book)
    if obs_a.ndim != cb_a.ndim:
        raise ValueError("Observation and code_book should have the same rank")
    obs_a = _as_2d(obs_a, "obs")
    cb_a = _as_2d(cb_a, "code_book")
    if obs_a.shape[1] != cb_a.shape[1]:
        raise ValueError("Code book(%d) and obs(%d) should have the same number of features "
                         "(e.g. columns)" % (cb_a.shape[1], obs_a.shape[1]))
    if cb_a.shape[0] < 1 or cb_a.shape[1] < 1:
        raise ValueError("empty codebook (shape %s)" % (cb_a.shape,))
    if obs_a.shape[1] > MAX_DIM:
        raise ValueError("vq: observations have %d features, this device search handles at most "
                         "%d" % (obs_a.shape[1], MAX_DIM))
    # scipy scans both arrays on the host first.  For a large observation matrix that scan costs
    # more than the whole device search, so it is replaced by a check of the result: with a finite
    # code book, a row holds an inf or NaN exactly when its distance is not finite.
    big = obs_a.size > (1 << 22)
    b = backend or default_backend()
    cbk = resident if resident is not None else b.codebook(cb_a)
    if check_finite:
        # the code book's verdict is remembered with its device copy (one scan per code book, not
        # one per call)
        cb_ok = getattr(cbk, "host_finite", None)
        if cb_ok is None:
            cb_ok = bool(np.isfinite(cb_a).all())
            try:
                cbk.host_finite = cb_ok
            except AttributeError:
                pass
        if not (cb_ok and (big or np.isfinite(obs_a).all())):
            raise ValueError("array must not contain infs or NaNs")
    ct = np.result_type(obs_a.dtype, cb_a.dtype)
    if ct.kind != "f":
        ct = np.dtype(np.float64)
    if obs_a.shape[0] == 0:
        return np.empty(0, dtype=np.int32), np.empty(0, dtype=ct)
    if hasattr(b, "assign_top1_host"):
        idx, dist = b.assign_top1_host(obs_a, cbk)
    else:
        idx, dist = b.assign_top1_np(obs_a, cbk, want_dist=True)
        idx, dist = idx.cpu().numpy(), dist.cpu().numpy()
    if check_finite and big and not np.isfinite(dist).all():
        raise ValueError("array must not contain infs or NaNs")
    return idx, dist.astype(ct)


def kmeans2(data, k, iter=10, thresh=1e-5, minit="points", missing="warn", check_finite=True,
            rng=None, seed=None, backend=None):
    """(codebook, label) with scipy.cluster.vq.kmeans2 semantics ('points' and 'matrix' inits).

    `thresh` is accepted and unused, as in scipy.
    """
    data = np.asarray(data)
    if check_finite and not np.isfinite(data).all():
        raise ValueError("array must not contain infs or NaNs")
    if rng is None:
        rng = seed
    return kmeans_fit(data, k, iters=iter, minit=minit, missing=missing, rng=rng, backend=backend)
