"""The library calls the reference itself makes on this path, for timing and cross-checks (TEST
INFRASTRUCTURE, like the rest of oracle/).

The reference's modules are Python 2 and import packages that are not installed, so they cannot
run; what CAN run are the third-party functions its call sites end in.  The functions below issue
exactly those calls with the reference's argument dtypes and loop structure, where scipy /
scikit-learn are installed (they return None otherwise).  bench.py times them as `cpu_baseline`.
"""
import numpy as np

from .descriptor_search import hist_normalise

__all__ = ["bow_quantize", "compute_vocab_features", "versions"]


def versions():
    out = {"numpy": np.__version__}
    for name in ("scipy", "sklearn"):
        try:
            out[name] = __import__(name).__version__
        except ImportError:
            out[name] = None
    return out


def bow_quantize(image_rows, vocab, norm="l1", power=0.5):
    """BagOfWords._quantize + _build_histogram (model/features.py:1162-1181): one scipy `vq` call
    per image with the image's float32 descriptors and the float64 vocabulary, then
    np.histogram(code, bins=range(0, K+1))[0] and the normalisation.  Returns (codes per image,
    histograms [n_img, K])."""
    from scipy.cluster.vq import vq
    vocab_size = len(vocab)
    codebooks = []
    for rows in image_rows:
        if rows is not None:
            code, _distortion = vq(rows, vocab)
            codebooks.append(code)
        else:
            codebooks.append([])
    hists = []
    for code in codebooks:
        hist = np.histogram(code, bins=range(0, vocab_size + 1))[0]
        hists.append(hist_normalise(hist, norm, power))
    return codebooks, np.asarray(hists)


def compute_vocab_features(vocab, feat):
    """vision/utils/scene_functions.py:375-392 with scikit-learn's euclidean_distances: argmin of
    the squared distances in batches of vocab.shape[0] rows.  None without scikit-learn."""
    try:
        from sklearn.metrics.pairwise import euclidean_distances
    except ImportError:
        return None
    batch_size = vocab.shape[0]
    k = feat.shape[0]
    codes = np.zeros(k, dtype=np.int64)
    for i in range(0, k, batch_size):
        sl = slice(i, i + batch_size)
        codes[sl] = np.argmin(euclidean_distances(feat[sl, :], vocab, squared=True), axis=1)
    return codes
