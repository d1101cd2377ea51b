"""k-means on the device: the assign step and the Lloyd loop.

The reference's own file of this name (model/clustering.py:19-177) is hierarchical clustering and
contains no k-means; the nearest-codeword assignment and Lloyd iterations this module provides are
what the reference reaches through scipy: `vq` at model/features.py:1166 and
`kmeans2(features, k, iter=iterations, minit='points')` at vision/facerec/face_indexing.py:156-161.
Semantics follow scipy's kmeans2 loop: per iteration assign (lowest index on ties), then the mean
of each cluster's members in float64, an empty cluster keeps its previous centroid, and the labels
returned are those of the last assignment, i.e. from before the last update.
"""
import warnings

import numpy as np
import torch

from .backend import default_backend

MAX_DIM = 2048   # widest row the descriptor search (and therefore the Lloyd loop) handles


def kmeans_assign(x, codebook, backend=None, return_dist=False):
    """Nearest-codeword index (int32) of every row of x; optionally the Euclidean distance."""
    b = backend or default_backend()
    cb = b.codebook(codebook)
    if isinstance(x, np.ndarray):
        idx, dist = b.assign_top1_np(x, cb, want_dist=return_dist)
    else:
        idx, dist = b.assign_top1(b.f32(x), cb, want_dist=return_dist)
    idx = idx.cpu().numpy()
    if return_dist:
        return idx, dist.cpu().numpy()
    return idx


def lloyd(xd, c0, iters, backend, allreduce=None, on_empty=None, labels0=None):
    """Run `iters` Lloyd iterations on device-resident rows.

    xd: float32 device tensor [n, d] (this process's rows); c0: float32 device tensor [k, d].
    allreduce(sums, counts): in-place sum over processes, or None for a single process.
    on_empty(): called once per iteration that left some cluster without members.
    labels0: labels from an earlier call with these rows (a guess for the first assignment).
    Returns (codebook device tensor, labels int32 device tensor, empties per iteration).
    """
    k, d = c0.shape
    # The search works on rows of a multiple of 128 columns and pads others with zeros on every
    # call; the rows are the same in every iteration, so they are padded once here.  Zero columns
    # change no distance, no sum and no mean (the centroids' padding stays zero), and the library
    # would have searched exactly these padded rows: same labels, same centroids.
    width = -(-d // 128) * 128
    if width != d and xd.shape[0] > 0:
        xd = torch.nn.functional.pad(xd, (0, width - d))
        c0 = torch.nn.functional.pad(c0, (0, width - d))
    cb = backend.make_codebook(c0)
    labels = labels0
    empties = []
    for _ in range(int(iters)):
        # the labels of the previous iteration are a guess of the new ones: same result, and the
        # search skips ranking candidates that cannot beat the guessed centroid
        labels, _dist = backend.assign_top1(xd, cb, want_dist=False, hint=labels)
        if hasattr(allreduce, "outputs"):
            # the collective's own buffer receives the sums: nothing is copied before the exchange
            sums, counts = backend.centroid_partials(xd, labels, k,
                                                     out=allreduce.outputs(k, c0.shape[1], xd.device))
        else:
            sums, counts = backend.centroid_partials(xd, labels, k)
        if allreduce is not None:
            allreduce(sums, counts)
        new_c, has = backend.centroid_finalize(sums, counts, cb.c)
        n_empty = int(k - int(has.sum().item()))
        empties.append(n_empty)
        if n_empty and on_empty is not None:
            on_empty()
        cb = backend.make_codebook(new_c)
    c = cb.c if width == d else cb.c[:, :d].contiguous()
    return c, labels, empties


def _warn_empty():
    warnings.warn("One of the clusters is empty. Re-run kmeans with a different initialization.",
                  stacklevel=3)


def kmeans_fit(x, k, iters=10, minit="points", missing="warn", rng=None, backend=None):
    """scipy.cluster.vq.kmeans2 semantics on the device.  Returns (codebook, labels) as numpy."""
    if int(iters) < 1:
        raise ValueError("Invalid iter (%s), must be a positive integer." % iters)
    if missing not in ("warn", "raise"):
        raise ValueError("Unknown missing method %r" % (missing,))
    x = np.asarray(x)
    if x.ndim == 1:
        x = x[:, None]
    if x.ndim != 2:
        raise ValueError("Input of rank > 2 is not supported.")
    if x.size < 1:
        raise ValueError("Empty input is not supported.")
    if x.shape[1] > MAX_DIM:
        raise ValueError("kmeans2: data has %d features, this device k-means handles at most %d"
                         % (x.shape[1], MAX_DIM))
    karr = np.asarray(k)
    if minit == "matrix" or karr.size > 1:
        init = np.array(karr, copy=True)
        if init.ndim == 1:
            init = init[:, None]
        if init.ndim != x.ndim:
            raise ValueError("k array doesn't match data rank")
        if init.shape[1] != x.shape[1]:
            raise ValueError("k array doesn't match data dimension")
    else:
        nc = int(karr)
        if nc < 1:
            raise ValueError("Cannot ask kmeans2 for %d clusters (k was %s)" % (nc, k))
        if minit != "points":
            raise ValueError("minit %r is not supported on the device; use 'points' or 'matrix'"
                             % (minit,))
        if nc > x.shape[0]:
            raise ValueError("Cannot take a larger sample than population when 'replace=False'")
        if rng is None or isinstance(rng, (int, np.integer)):
            rng = np.random.default_rng(rng)
        init = x[rng.choice(x.shape[0], size=nc, replace=False)]

    def on_empty():
        if missing == "raise":
            raise RuntimeError("One of the clusters is empty. "
                               "Re-run kmeans with a different initialization.")
        _warn_empty()

    b = backend or default_backend()
    xd = b.f32(x)
    c, labels, _ = lloyd(xd, b.f32(init), iters, b, on_empty=on_empty)
    out_dtype = x.dtype if x.dtype.kind == "f" else np.float64
    return c.cpu().numpy().astype(out_dtype), labels.cpu().numpy()


# ---- histogram comparison functions of the reference's clustering module -------------------------
def _hist_compare(p1, p2, metric, backend=None):
    """One value of a histogram metric between two 1-D histograms (model/clustering.py:313-320:
    both are converted to float32, the value comes back as a Python float)."""
    p1 = np.asarray(p1)
    p2 = np.asarray(p2)
    if p1.ndim != 1 or p1.shape != p2.shape:
        raise ValueError("expected two 1-D histograms of the same length, got %s and %s"
                         % (p1.shape, p2.shape))
    b = backend or default_backend()
    return float(b.pairwise(b.f32(p1[None, :]), b.f32(p2[None, :]), metric).cpu().numpy()[0, 0])


def chisqr(p1, p2):
    """Chi-square histogram distance sum((p1-p2)^2 / p1) (model/clustering.py:305-306).
    `KNNScikit(metric=chisqr)` recognises this function and runs the batched device kernel."""
    return _hist_compare(p1, p2, "chisqr")


def intersection(p1, p2):
    """Histogram intersection sum(min(p1, p2)) (model/clustering.py:309-310).
    `KNNScikit(metric=intersection)` recognises this function and runs the batched device
    kernel; like the reference, the neighbour search treats the value as a distance."""
    return _hist_compare(p1, p2, "intersection")
