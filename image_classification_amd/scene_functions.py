"""`compute_vocab_features` and `compute_spatial_features` with the reference's signatures.

Mirrors vision/utils/scene_functions.py:375-392: `codes[i] = argmin_j |feat_i - vocab_j|^2` as an
int64 array (the reference allocates `np.zeros(k, dtype=np.int)`), first minimum on ties.  The
reference walks the rows in batches of `vocab.shape[0]` through sklearn's euclidean_distances to
bound host memory; the device kernel never materialises the distance matrix, so one call covers
all rows.
"""
import numpy as np

from .backend import default_backend

__all__ = ["compute_vocab_features", "compute_spatial_features", "spatial_cell_weights",
           "spatial_keypoint_cells", "compute_spatial_features_from_xml"]


def compute_vocab_features(vocab, feat, backend=None):
    vocab = np.asarray(vocab)
    feat = np.asarray(feat)
    if vocab.ndim != 2 or feat.ndim != 2:
        raise ValueError("vocab and feat must be 2-D")
    if vocab.shape[1] != feat.shape[1]:
        raise ValueError("Incompatible dimension for X and Y matrices: X.shape[1] == %d while "
                         "Y.shape[1] == %d" % (feat.shape[1], vocab.shape[1]))
    if feat.shape[1] > 2048:
        raise ValueError("compute_vocab_features: %d features per row, this device search handles "
                         "at most 2048" % feat.shape[1])
    if feat.shape[0] == 0:
        return np.zeros(0, dtype=np.int64)
    b = backend or default_backend()
    idx, _ = b.assign_top1_np(feat, b.codebook(vocab), want_dist=False)
    return idx.cpu().numpy().astype(np.int64)


def spatial_cell_weights(num_levels):
    """Weights of the pyramid cells, level 0 first (vision/utils/scene_functions.py:357-358):
    2^(1-L) for the whole image, then 2^-i for the 4^(L-i) cells of level L-i, i = L-1 .. 1."""
    w = [pow(2.0, 1 - num_levels)]
    for i in range(num_levels - 1, 0, -1):
        w += [pow(2.0, -i)] * pow(4, num_levels - i)
    return np.asarray(w, dtype=np.float64)


def spatial_keypoint_cells(keypoints, num_levels, im_width, im_height):
    """Cell number of every keypoint on every level: int32 [num_levels, n_kp].

    Same arithmetic as vision/utils/scene_functions.py:293-305: keypoints are (col, row); level 0
    is cell 0; on level l the grid position is (int(row / (im_height / 2^l)), int(col /
    (im_width / 2^l))) and the cell is base_l + 2^l * grid_col + grid_row (column-major), with
    base_l = 1 + 4 + ... + 4^(l-1).  A keypoint outside the image gets -1 (no vote); the
    reference would raise an IndexError or vote into a wrong cell there.
    """
    kp = np.asarray(keypoints, dtype=np.float64).reshape(-1, 2)
    cells = np.zeros((num_levels, kp.shape[0]), dtype=np.int32)
    base = 0
    for level in range(1, num_levels):
        base += 4 ** (level - 1)
        row_div = im_height / (2. ** level)
        col_div = im_width / (2. ** level)
        g_row = np.trunc(kp[:, 1] / row_div).astype(np.int64)
        g_col = np.trunc(kp[:, 0] / col_div).astype(np.int64)
        cell = base + (2 ** level) * g_col + g_row
        side = 2 ** level
        ok = (g_row >= 0) & (g_row < side) & (g_col >= 0) & (g_col < side)
        cells[level] = np.where(ok, cell, -1)
    return cells


def compute_spatial_features(vocab_feat, keypoints, vocab_size, num_levels, im_width, im_height,
                             backend=None):
    """Spatial pyramid features, float64 [NUM_IMAGES, n_cells * vocab_size].

    Mirrors vision/utils/scene_functions.py:338-373 (and its helper :279-307): `vocab_feat` holds
    the codes [NUM_IMAGES, NUM_KEYPOINTS], `keypoints` [NUM_KEYPOINTS, 2] as (col, row) are shared
    by all images.  The reference loops over keypoints in Python; here one device call counts all
    images and applies the cell weights.
    """
    vocab_feat = np.asarray(vocab_feat)
    if vocab_feat.ndim != 2:
        raise ValueError("vocab_feat must be [NUM_IMAGES, NUM_KEYPOINTS]")
    keypoints = np.asarray(keypoints)
    if keypoints.shape[0] != vocab_feat.shape[1]:
        raise ValueError("one keypoint per column of vocab_feat is required")
    weights = spatial_cell_weights(num_levels)
    cells = spatial_keypoint_cells(keypoints, num_levels, im_width, im_height)
    if vocab_feat.shape[0] == 0:
        return np.zeros((0, len(weights) * vocab_size))
    b = backend or default_backend()
    out = b.spatial_hist(b.i32(vocab_feat.astype(np.int32)), b.i32(cells), b.f64(weights),
                         int(vocab_size))
    return out.cpu().numpy()


def compute_spatial_features_from_xml(extract_path, projected_path, vocab_path, num_levels,
                                      backend=None):
    """Spatial-pyramid features straight from the XML / .desc outputs of the vision binaries
    (vision/utils/scene_functions.py:309-336): the vocabulary and the projected descriptors are raw
    float32 files next to their XML headers, the keypoint grid and the image geometry come from
    the extraction XML.  The descriptor file is memory-mapped, so it streams to the device without
    a second host copy."""
    from . import bovw_files as bf
    dims = bf.extract_from_xml(vocab_path, ["numCenters", "dimension"])
    vocab = bf.binary_to_ndarray("%s.desc" % vocab_path, dims, np.float32)
    dims = bf.extract_from_xml(projected_path, ["nDescriptors", "cols"])
    proj = np.memmap("%s.desc" % projected_path, dtype=np.float32, mode="r", shape=tuple(dims))
    keypoints = bf.load_opencv_matrix(extract_path, "keypoints").squeeze()
    im_height, im_width, num_images = bf.extract_from_xml(
        extract_path, ["image_height", "image_width", "nImg"])
    vocab_feat = compute_vocab_features(vocab, proj, backend=backend).reshape(
        num_images, keypoints.shape[0])
    return compute_spatial_features(vocab_feat, keypoints, vocab.shape[0], num_levels, im_width,
                                    im_height, backend=backend)
