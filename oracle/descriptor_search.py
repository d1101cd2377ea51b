"""float64 numpy restatement of the reference's descriptor-search arithmetic (test oracle)."""
import numpy as np

__all__ = ["sq_distances", "vq", "compute_vocab_features", "topk", "kneighbors",
           "ratio_matches", "robust_match_descriptors", "build_histogram", "hist_normalise",
           "update_cluster_means", "kmeans2_matrix", "top2_merge", "compute_spatial_features",
           "pairwise_metric", "kneighbors_metric", "knn_predict_proba"]


def sq_distances(x, c):
    """|x_i|^2 - 2 x_i.c_j + |c_j|^2 in float64, clamped at 0.

    The expansion both back ends use: scipy's _vq.vq (BLAS path) behind model/features.py:1166 and
    sklearn's euclidean_distances behind vision/utils/scene_functions.py:391.
    """
    x = np.asarray(x, dtype=np.float64)
    c = np.asarray(c, dtype=np.float64)
    d2 = (x * x).sum(1)[:, None] - 2.0 * (x @ c.T) + (c * c).sum(1)[None, :]
    np.maximum(d2, 0.0, out=d2)
    return d2


def _batches(n, k):
    step = max(1, min(n, (1 << 25) // max(1, k)))
    for lo in range(0, n, step):
        yield lo, min(n, lo + step)


def vq(obs, code_book):
    """scipy.cluster.vq.vq as called at model/features.py:1166 / face_indexing.py:141.

    code int32 = first minimum over the codebook (scipy scans ascending with '<'), dist =
    sqrt(max(d^2, 0)) in the promoted dtype (float64 here).
    """
    obs = np.asarray(obs, dtype=np.float64)
    code_book = np.asarray(code_book, dtype=np.float64)
    n = obs.shape[0]
    code = np.empty(n, dtype=np.int32)
    dist = np.empty(n, dtype=np.float64)
    for lo, hi in _batches(n, code_book.shape[0]):
        d2 = sq_distances(obs[lo:hi], code_book)
        j = d2.argmin(1)
        code[lo:hi] = j
        dist[lo:hi] = np.sqrt(d2[np.arange(hi - lo), j])
    return code, dist


def compute_vocab_features(vocab, feat):
    """vision/utils/scene_functions.py:375-392: argmin of squared distances in batches of
    vocab.shape[0] rows, int64 codes."""
    batch = vocab.shape[0]
    k = feat.shape[0]
    codes = np.zeros(k, dtype=np.int64)
    for i in range(0, k, batch):
        sl = slice(i, i + batch)
        codes[sl] = np.argmin(sq_distances(feat[sl, :], vocab), axis=1)
    return codes


def topk(x, index, kk=3):
    """The kk nearest index rows per row of x by (distance, index): (dist [n,kk] Euclidean
    float64, ind [n,kk] int64).  Distances of the winners are recomputed directly as
    sqrt(sum (x-c)^2), free of the expansion's cancellation.  Missing neighbours (index smaller
    than kk) are (inf, -1)."""
    x = np.asarray(x, dtype=np.float64)
    index = np.asarray(index, dtype=np.float64)
    n, k = x.shape[0], index.shape[0]
    take = min(kk, k)
    dist = np.full((n, kk), np.inf)
    ind = np.full((n, kk), -1, dtype=np.int64)
    for lo, hi in _batches(n, k):
        d2 = sq_distances(x[lo:hi], index)
        if take < k:
            part = np.argpartition(d2, take - 1, axis=1)[:, :take]
        else:
            part = np.tile(np.arange(k), (hi - lo, 1))
        rows = np.arange(hi - lo)[:, None]
        # exact distances of the candidates, then order by (distance, index)
        diff = x[lo:hi, None, :] - index[part]
        dd = np.sqrt((diff * diff).sum(2))
        order = np.lexsort((part, dd), axis=1)
        dist[lo:hi, :take] = dd[rows, order]
        ind[lo:hi, :take] = part[rows, order]
    return dist, ind


def kneighbors(fit_x, x, n_neighbors=2):
    """KNeighborsClassifier(algorithm='brute').fit(fit_x).kneighbors(x, n_neighbors)
    (model/classifiers.py:195-197 <- model/robust_matching.py:80): ascending Euclidean
    distances float64 [M,k], indices int64 [M,k]."""
    dist, ind = topk(x, fit_x, n_neighbors)
    return dist, ind


def ratio_matches(dist, ind, thresh):
    """The loop at model/robust_matching.py:81-84: rows idx with d1 < d2 * thresh.
    Returns (matches_q = ind[idx, 0], matches_im = idx)."""
    keep = dist[:, 0] < dist[:, 1] * thresh
    return ind[keep, 0].astype(np.int64), np.nonzero(keep)[0].astype(np.int64)


def robust_match_descriptors(query_desc, cand_descs, min_points, thresh):
    """The k-NN stage of RobustMatcher.match (model/robust_matching.py:68-84) per candidate."""
    out = []
    for d in cand_descs:
        rows = 0 if d is None else d.shape[0]
        if rows > min_points:
            dist, ind = kneighbors(query_desc, d, 2)
            out.append(ratio_matches(dist, ind, thresh))
        else:
            out.append((np.empty(0, dtype=np.int64), np.empty(0, dtype=np.int64)))
    return out


def hist_normalise(hist, norm="l1", power=0.5):
    """'l1' is the reference's rule hist * (1.0 / (s or 1)) (model/features.py:1178-1179).
    'l2' and 'power_l2' are this build's extensions (no counterpart in the reference)."""
    hist = np.asarray(hist)
    if norm == "none":
        return hist.astype(np.float64)
    if norm == "l1":
        s = hist.sum()
        return hist * (1.00 / (s or 1))
    y = hist.astype(np.float64)
    if norm == "power_l2":
        y = np.sqrt(y) if power == 0.5 else np.where(y == 0, 0.0, np.power(y, power))
    elif norm != "l2":
        raise ValueError(norm)
    nrm = np.sqrt((y * y).sum())
    return y / (nrm or 1.0)


def build_histogram(codebooks, vocab_size, norm="l1", power=0.5):
    """BagOfWords._build_histogram (model/features.py:1174-1181): per image
    np.histogram(code, bins=range(0, K+1))[0], then normalisation; float64 [n_img, K]."""
    hists = []
    for code in codebooks:
        hist = np.histogram(code, bins=range(0, vocab_size + 1))[0]
        hists.append(hist_normalise(hist, norm, power))
    return np.asarray(hists)


def update_cluster_means(data, label, nc):
    """scipy _vq.update_cluster_means: float64 member means; has_members marks non-empty."""
    data = np.asarray(data, dtype=np.float64)
    sums = np.zeros((nc, data.shape[1]))
    np.add.at(sums, label, data)
    counts = np.bincount(label, minlength=nc)
    has = counts > 0
    cb = np.zeros_like(sums)
    cb[has] = sums[has] / counts[has][:, None]
    return cb, has, sums, counts


def kmeans2_matrix(data, init, iters):
    """scipy.cluster.vq.kmeans2(data, init, iter=iters, minit='matrix', missing='warn') as used
    (with minit='points') at vision/facerec/face_indexing.py:158.  Per iteration: vq assign,
    member means, empty clusters keep their previous centroid.  Returns (codebook, labels of the
    last assignment, [codebook after each iteration])."""
    data = np.asarray(data, dtype=np.float64)
    cb = np.array(init, dtype=np.float64, copy=True)
    nc = cb.shape[0]
    history = []
    label = None
    for _ in range(iters):
        label = vq(data, cb)[0]
        new_cb, has, _, _ = update_cluster_means(data, label, nc)
        new_cb[~has] = cb[~has]
        cb = new_cb
        history.append(cb.copy())
    return cb, label, history


def top2_merge(shard_results, shard_base):
    """Merge per-shard (dist [n,2], ind [n,2]) by (distance, global index)."""
    dist = np.concatenate([d for d, _ in shard_results], axis=1)
    ind = np.concatenate([np.where(i >= 0, i + b, -1) for (_, i), b in zip(shard_results, shard_base)],
                         axis=1)
    key_i = np.where(ind >= 0, ind, np.iinfo(np.int64).max)
    order = np.lexsort((key_i, dist), axis=1)[:, :2]
    rows = np.arange(dist.shape[0])[:, None]
    return dist[rows, order], ind[rows, order]


def _get_spatial_histogram(vocab_feat, keypoints, vocab_size, num_levels, im_width, im_height,
                           num_cells):
    """vision/utils/scene_functions.py:279-307, statement by statement (a Python loop over the
    keypoints; level 0 is cell 0, deeper cells are numbered column-major after a running base)."""
    num_images = vocab_feat.shape[0]
    counts_mat = np.zeros((num_images, num_cells, vocab_size))
    image_range = np.arange(num_images)
    for kp_id, kp in enumerate(keypoints):
        vocab_for_kp = vocab_feat[:, kp_id]
        col, row = kp[0], kp[1]
        counts_mat[image_range, 0, vocab_for_kp] += 1
        base_cell_count = 0
        for level in range(1, num_levels):
            base_cell_count += 4 ** (level - 1)
            row_div = im_height / (2. ** level)
            col_div = im_width / (2. ** level)
            grid_pos = (int(row / row_div), int(col / col_div))
            cell_id = base_cell_count + (2 ** level) * grid_pos[1] + grid_pos[0]
            counts_mat[image_range, cell_id, vocab_for_kp] += 1
    return counts_mat


def compute_spatial_features(vocab_feat, keypoints, vocab_size, num_levels, im_width, im_height):
    """vision/utils/scene_functions.py:338-373: weighted pyramid counts, one row per image."""
    cell_weights = np.asarray([pow(2, 1 - num_levels)] +
                              sum([[pow(2, -i)] * pow(4, num_levels - i)
                                   for i in range(num_levels - 1, 0, -1)], []))
    num_cells = len(cell_weights)
    num_images = vocab_feat.shape[0]
    features = np.zeros((num_images, num_cells * vocab_size))
    feature_mat = _get_spatial_histogram(vocab_feat, keypoints, vocab_size, num_levels, im_width,
                                         im_height, num_cells)
    for image_id, mat in enumerate(feature_mat):
        weighted_mat = mat * cell_weights[:, np.newaxis]
        features[image_id, :] = np.hstack(weighted_mat)
    return features


def pairwise_metric(x, c, metric="manhattan"):
    """Distance matrix [m, n] in float64 between float32 rows under the histogram-space metrics
    of KNNScikit (model/classifiers.py:170-172): 'euclidean', 'sqeuclidean', 'manhattan'
    (vision/logo_recognition/training.py:41), 'intersection' = sum min(a, b) and 'chisqr' =
    sum (a-b)^2 / a over bins with a != 0 (OpenCV compareHist, model/clustering.py:305-320; the
    reference converts both rows to float32 first, :318-319).  Row by row: no [m, n, d] cube."""
    x = np.asarray(x, dtype=np.float32).astype(np.float64)
    c = np.asarray(c, dtype=np.float32).astype(np.float64)
    out = np.empty((x.shape[0], c.shape[0]), dtype=np.float64)
    for i in range(x.shape[0]):
        a = x[i][None, :]
        if metric in ("manhattan", "cityblock", "l1"):
            out[i] = np.abs(a - c).sum(1)
        elif metric == "intersection":
            out[i] = np.minimum(a, c).sum(1)
        elif metric == "chisqr":
            keep = np.abs(x[i]) > np.finfo(np.float64).eps
            t = a[:, keep] - c[:, keep]
            out[i] = (t * t / a[:, keep]).sum(1)
        elif metric in ("euclidean", "l2"):
            out[i] = np.sqrt(((a - c) ** 2).sum(1))
        elif metric == "sqeuclidean":
            out[i] = ((a - c) ** 2).sum(1)
        else:
            raise ValueError("unknown metric %r" % (metric,))
    return out


def kneighbors_metric(fit_x, x, n_neighbors, metric="manhattan"):
    """KNeighborsClassifier(metric=...).fit(fit_x).kneighbors(x, n_neighbors): the n_neighbors
    smallest entries of every row of the distance matrix, ascending; ties by index (stable)."""
    dist = pairwise_metric(x, fit_x, metric)
    ind = np.argsort(dist, axis=1, kind="stable")[:, :n_neighbors]
    return np.take_along_axis(dist, ind, axis=1), ind.astype(np.int64)


def knn_predict_proba(neigh_dist, neigh_ind, labels, classes, weights="distance"):
    """KNeighborsClassifier.predict_proba from the neighbour lists (model/classifiers.py:183-185):
    every neighbour votes for its label with weight 1/distance ('distance') or 1 ('uniform'); a
    row with zero-distance neighbours counts only those; rows are normalised to sum 1."""
    labels = np.asarray(labels)
    classes = list(classes)
    w = np.ones_like(neigh_dist, dtype=np.float64)
    if weights == "distance":
        with np.errstate(divide="ignore"):
            w = 1.0 / neigh_dist
        zero = np.isinf(w)
        rows = zero.any(axis=1)
        w[rows] = zero[rows].astype(np.float64)
    proba = np.zeros((neigh_dist.shape[0], len(classes)), dtype=np.float64)
    col = np.searchsorted(np.asarray(classes), labels[neigh_ind])
    for j in range(neigh_ind.shape[1]):
        np.add.at(proba, (np.arange(proba.shape[0]), col[:, j]), w[:, j])
    s = proba.sum(axis=1, keepdims=True)
    s[s == 0] = 1.0
    return proba / s
