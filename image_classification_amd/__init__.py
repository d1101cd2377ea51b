"""MI355X-native nearest-codeword / nearest-neighbour search over local image descriptors.

A drop-in for one path of winkash/image-classification: the calls into scipy.cluster.vq and
scikit-learn's brute-force k-NN made by BagOfWords quantisation (model/features.py:1162-1181),
compute_vocab_features (vision/utils/scene_functions.py:375-392), the RobustMatcher ratio test
(model/robust_matching.py:68-84) and the kmeans2 codebook training
(vision/facerec/face_indexing.py:156-161).  Python keeps the reference's numpy-in / numpy-out
signatures; the arithmetic runs in libic_hip.so (hand-written gfx950 kernels behind the C ABI of
include/ic_hip.h).  Importing this package touches neither HIP nor the shared library.
"""
from .imagedesc import ImageDesc
from .vq import vq, kmeans2
from .clustering import chisqr, intersection, kmeans_assign, kmeans_fit
from .features import BagOfWords, SpatialBagOfWords, Vocabulary
from .scene_functions import (compute_spatial_features, compute_spatial_features_from_xml,
                              compute_vocab_features)
from .classifiers import KNNScikit
from .robust_matching import RobustMatcher
from .shortlist import KnnTrainer, fit_shortlist_index

__all__ = ["ImageDesc", "vq", "kmeans2", "kmeans_assign", "kmeans_fit", "BagOfWords",
           "SpatialBagOfWords", "Vocabulary", "compute_vocab_features", "compute_spatial_features",
           "compute_spatial_features_from_xml", "KNNScikit", "KnnTrainer", "fit_shortlist_index", "RobustMatcher",
           "intersection", "chisqr"]
__version__ = "0.1.0"
