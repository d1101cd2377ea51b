"""Bag-of-words quantisation with the reference's class and method names.

Mirrors `BagOfWords` (model/features.py:1125-1284) and `Vocabulary` (model/features.py:1329-1391):
same constructor arguments, `train` / `extract` return `(image_descs, hists)` with `hists` a
float64 array [n_images, vocabsize], images whose `descriptors` is None give an all-zero row, and
the histogram is L1-normalised exactly as `hist * (1.0 / (sum or 1))` (model/features.py:1174-1181).

What differs from the reference is only how the work is issued: the reference calls scipy's `vq`
once per image in a Python loop (model/features.py:1164-1169) and builds each histogram with
np.histogram; here all descriptors of a call are concatenated, one device call assigns them and
one device call builds every histogram.

`norm` is an extension: 'l1' is the reference's normalisation, 'l2', 'power_l2' and 'none' are
additional modes with no counterpart in the reference.
"""
import json
import os

import numpy as np

from .backend import default_backend
from .clustering import kmeans_fit
from .imagedesc import ImageDesc

__all__ = ["BagOfWords", "SpatialBagOfWords", "Vocabulary", "register_component"]

# Classes a model directory's manifest may name.  A manifest is data from disk: it can only select
# among classes the running program has registered (or handed to load_from_dir), never make the
# loader import a module of its choosing.
_COMPONENTS = {}


def register_component(klass):
    """Class decorator: allow `klass` (a feature extractor or projection with `save_to_file` /
    `load_from_file`) to be named in a model directory's manifest."""
    _COMPONENTS[_import_path(klass)] = klass
    return klass


class Vocabulary(object):
    """Vocabulary trainer.

    The reference writes the descriptors to an OpenCV-XML + raw float32 file pair and runs an
    external `vocab_kms` binary whose source is not part of the reference tree
    (model/features.py:1367-1378, vision/utils/bovw_functions.py:111-119).  This class trains on
    the device instead, with the Lloyd semantics of the only k-means whose code the reference
    does show (scipy kmeans2, minit='points'; vision/facerec/face_indexing.py:158).  Like the
    reference's `_read_vocab` (model/features.py:1335-1341) it returns a float64 array.
    """

    def __init__(self, vocab_size, iterations=20, seed=None, backend=None):
        self.vocab_size = vocab_size
        self.iterations = iterations
        self.seed = seed
        self._backend = backend
        self.vocab = None

    def train(self, feats):
        feats = np.asarray(feats)
        # same guard as vocab_kms's wrapper (vision/utils/bovw_functions.py:116-117)
        assert self.vocab_size <= feats.shape[0], \
            "Number of clusters should be less than number of descriptors"
        self.dim = feats.shape[1]
        centers, _ = kmeans_fit(feats.astype(np.float32, copy=False), self.vocab_size,
                                iters=self.iterations, minit="points", rng=self.seed,
                                backend=self._backend)
        self.vocab = centers.astype(np.float64)
        return self.vocab

    def _write_to_file(self, feats):
        """Descriptor file pair in a temporary location (model/features.py:1357-1365)."""
        from tempfile import mkstemp
        from .bovw_files import write_descriptor_file
        h, feat_file = mkstemp()
        os.close(h)
        os.unlink(feat_file)
        return write_descriptor_file(feat_file + ".xml", feats)

    def _read_vocab(self, vocab_file):
        """float64 [vocab_size, dim] from `<vocab_file>.desc` (model/features.py:1335-1341)."""
        from .bovw_files import binary_to_ndarray
        return binary_to_ndarray(vocab_file + ".desc", (self.vocab_size, self.dim),
                                 np.float32).astype(np.float64)

    def save_to_file(self, file_path):
        assert self.vocab is not None, "Vocabulary is missing"
        with open(file_path, "wb") as f:
            np.save(f, self.vocab)

    @classmethod
    def load_from_file(cls, file_path):
        with open(file_path, "rb") as f:
            vocab = np.load(f)
        obj = cls(len(vocab))
        obj.vocab = vocab
        return obj


class BagOfWords(object):
    """Bag-of-words quantiser with the reference's constructor, `train` and `extract`
    (model/features.py:1125-1212).

    Model directories are this package's own design, not the reference's pickled class table
    (model/features.py:1218-1268): `save_to_dir` writes `bow.json` (a manifest: format version,
    normalisation, the import paths of the extractor / projection classes, any subclass
    parameters) next to `vocab.npy` -- the one file both layouts share -- and lets the extractor
    and the projection write their own files through their `save_to_file`.  `load_from_dir`
    looks the class names of the manifest up among the classes registered with
    `register_component` (or passed as `components=`) and calls their `load_from_file`; nothing
    is unpickled and no module is imported on the manifest's say-so.  A directory written by the
    reference (no manifest) still yields its vocabulary: pass the extractor (and projection)
    objects to `load_from_dir` in that case.
    """
    MANIFEST = "bow.json"
    VOCAB_FILE = "vocab.npy"
    _component_files = {"feat_ext": "feat_ext", "projection": "project"}
    _extra_params = ()          # attribute names a subclass wants in the manifest

    def __init__(self, feat_ext, vocabsize=None, vocab=None, projection=None, norm="l1",
                 power=0.5, backend=None):
        assert hasattr(feat_ext, "extract")
        if projection:
            assert hasattr(projection, "train") and hasattr(projection, "project")
        if vocab is None:
            assert vocabsize, "Vocabsize required for training"
        self.feat_ext = feat_ext
        self.projection = projection
        self.vocab = vocab
        self.vocabsize = len(vocab) if vocab is not None else vocabsize
        self.norm = norm
        self.power = power
        self._backend = backend

    def _b(self):
        return getattr(self, "_backend", None) or default_backend()

    # The vocabulary is held as a private read-only copy: nothing outside can edit it in place,
    # so its device copy is made once and reused without hashing ten megabytes per call (a
    # per-image `extract` otherwise spends half its time proving that the array is unchanged).
    # Assigning `bow.vocab = other` replaces it; `bow.vocab[...] = x` raises.
    @property
    def vocab(self):
        held = getattr(self, "_vocab", None)
        return None if held is None else held.view()

    @vocab.setter
    def vocab(self, value):
        self._vocab_cb = None
        if value is None:
            self._vocab = None
            return
        held = np.array(value, copy=True)
        held.setflags(write=False)
        self._vocab = held

    def __getstate__(self):
        state = dict(self.__dict__)
        state["_vocab_cb"] = None        # device memory does not travel
        return state

    def _codebook(self):
        """The backend's resident form of the vocabulary, made once per (process, backend)."""
        b = self._b()
        key = (os.getpid(), id(b))
        got = getattr(self, "_vocab_cb", None)
        if got is None or got[0] != key:
            got = (key, b.codebook(self._vocab))
            self._vocab_cb = got
        return got[1]

    # -- descriptors -----------------------------------------------------------------------------
    # The reference's helper names (model/features.py:1141-1190), kept because subclasses hook
    # into them; `train` / `extract` go through them.
    def _feature_extract(self, image_list):
        return [ImageDesc(*r) for r in self.feat_ext.extract(image_list)]

    def _merge_descs_to_feats(self, image_descs):
        return np.concatenate([d.descriptors for d in image_descs if d.descriptors is not None])

    def _learn_projection(self, all_feats):
        self.projection.train(all_feats)

    def _project(self, image_descs):
        return [ImageDesc(d.cv_keypoints,
                          None if d.descriptors is None else self.projection.project(d.descriptors))
                for d in image_descs]

    def _describe(self, image_list, fit_projection=False):
        """Extractor output as ImageDescs, projected when a projection is configured (fitted on
        all descriptors of the call first when `fit_projection`)."""
        descs = self._feature_extract(image_list)
        if not self.projection:
            return descs
        if fit_projection:
            self._learn_projection(self._merge_descs_to_feats(descs))
        return self._project(descs)

    def _stack(self, image_descs):
        return self._merge_descs_to_feats(image_descs)

    def _learn_vocab(self, all_feats):
        self._vocab_trainer = Vocabulary(self.vocabsize, backend=getattr(self, "_backend", None))
        self.vocab = self._vocab_trainer.train(all_feats)
        return self.vocab

    # -- the hot path ----------------------------------------------------------------------------
    @staticmethod
    def _parts_and_offsets(image_descs):
        """The non-empty descriptor arrays in image order and every image's row range."""
        rows = [0]
        parts = []
        for idesc in image_descs:
            n = 0
            if idesc.descriptors is not None:
                n = idesc.descriptors.shape[0]
                if n:
                    parts.append(np.asarray(idesc.descriptors))
            rows.append(rows[-1] + n)
        return parts, np.asarray(rows, dtype=np.int64)

    def _assign_all(self, image_descs):
        """One device call for every descriptor of every image.

        Returns (codes int32 device tensor [total], offsets int64 numpy [n_img + 1]).
        """
        b = self._b()
        parts, offsets = self._parts_and_offsets(image_descs)
        if not parts:
            return b.i32(np.empty(0, dtype=np.int32)), offsets
        # no host-side concatenation: the per-image arrays are gathered into the upload buffer
        codes, _ = b.assign_top1_parts(parts, self._codebook(), want_dist=False)
        return codes, offsets

    def _quantize(self, image_descs):
        if type(self)._build_histogram is BagOfWords._build_histogram:
            b = self._b()
            if hasattr(b, "quantize_parts"):
                # search, histograms and their way back to the host as one pipeline
                parts, offsets = self._parts_and_offsets(image_descs)
                _codes, hists = b.quantize_parts(parts, offsets, self._codebook(),
                                                 len(self.vocab), self.norm, self.power)
                return image_descs, hists
            codes, offsets = self._assign_all(image_descs)
            hists = b.hist(codes, b.i64(offsets), len(self.vocab), self.norm, self.power)
            return image_descs, b.to_host(hists)
        codes, offsets = self._assign_all(image_descs)
        # a subclass builds its own histogram from per-image code lists (SpatialBagOfWords does,
        # model/features.py:1298-1308): hand it what the reference hands it
        host = codes.cpu().numpy()
        codebooks = []
        for i, idesc in enumerate(image_descs):
            if idesc.descriptors is not None:
                codebooks.append(host[offsets[i]:offsets[i + 1]])
            else:
                codebooks.append([])
        return image_descs, self._build_histogram(image_descs, codebooks)

    def _build_histogram(self, image_descs, codebooks):
        b = self._b()
        rows = [0]
        parts = []
        for code in codebooks:
            code = np.asarray(code, dtype=np.int32).reshape(-1)
            parts.append(code)
            rows.append(rows[-1] + code.shape[0])
        allc = np.concatenate(parts) if parts else np.empty(0, dtype=np.int32)
        if len(codebooks) == 0:
            return np.asarray([])
        hists = b.hist(b.i32(allc), b.i64(np.asarray(rows, dtype=np.int64)), len(self.vocab),
                       self.norm, self.power)
        return hists.cpu().numpy()

    def train(self, image_list):
        """Fit the projection (if any) and the vocabulary (if none was given) on the images, then
        quantise them: (image_descs, hists), as model/features.py:1192-1204 returns."""
        image_descs = self._describe(image_list, fit_projection=True)
        if self.vocab is None:
            self._learn_vocab(self._stack(image_descs))
        return self._quantize(image_descs)

    def extract(self, image_list):
        """(image_descs, hists float64 [n_images, vocabsize]) as model/features.py:1206-1212."""
        return self._quantize(self._describe(image_list))

    # -- model directory -------------------------------------------------------------------------
    def save_to_dir(self, model_dir):
        assert self.vocab is not None, "vocab is not set"
        os.makedirs(model_dir, exist_ok=True)
        manifest = {"format": 1, "class": _import_path(type(self)), "norm": self.norm,
                    "power": self.power, "components": {},
                    "params": {name: getattr(self, name) for name in self._extra_params}}
        for attr, fname in self._component_files.items():
            obj = getattr(self, attr)
            if obj:
                obj.save_to_file(os.path.join(model_dir, fname))
                manifest["components"][attr] = _import_path(type(obj))
        np.save(os.path.join(model_dir, self.VOCAB_FILE), np.asarray(self.vocab))
        with open(os.path.join(model_dir, self.MANIFEST), "w") as f:
            json.dump(manifest, f, indent=1, sort_keys=True)

    @classmethod
    def load_from_dir(cls, model_dir, feat_ext=None, projection=None, backend=None,
                      components=None):
        """`components`: classes (a list, or a mapping 'module:Class' -> class) the manifest may
        name in addition to those registered with `register_component`."""
        vocab = np.load(os.path.join(model_dir, cls.VOCAB_FILE), allow_pickle=False)
        manifest = {"format": 1, "norm": "l1", "power": 0.5, "components": {}, "params": {}}
        path = os.path.join(model_dir, cls.MANIFEST)
        if os.path.exists(path):
            with open(path) as f:
                manifest.update(json.load(f))
        if manifest["format"] != 1:
            raise ValueError("%s: manifest format %r is not supported (this package reads format 1)"
                             % (path, manifest["format"]))
        allowed = dict(_COMPONENTS)
        if components:
            if not hasattr(components, "items"):
                components = {_import_path(k): k for k in components}
            allowed.update(components)
        given = {"feat_ext": feat_ext, "projection": projection}
        for attr, fname in cls._component_files.items():
            if given[attr] is None and attr in manifest["components"]:
                name = manifest["components"][attr]
                if name not in allowed:
                    raise ValueError(
                        "%s names %s class %r, which is not registered: decorate it with "
                        "register_component, or pass components=[...] / %s= to load_from_dir"
                        % (path, attr, name, attr))
                given[attr] = allowed[name].load_from_file(os.path.join(model_dir, fname))
        if given["feat_ext"] is None:
            raise ValueError("%s names no feature extractor: pass feat_ext=" % model_dir)
        obj = cls.__new__(cls)
        BagOfWords.__init__(obj, given["feat_ext"], vocab=vocab, projection=given["projection"],
                            norm=manifest["norm"], power=manifest["power"], backend=backend)
        for name in cls._extra_params:
            setattr(obj, name, manifest["params"][name])
        return obj


def _import_path(klass):
    return "%s:%s" % (klass.__module__, klass.__qualname__)



class SpatialBagOfWords(BagOfWords):
    """Bag of words with a spatial pyramid (model/features.py:1287-1326): every image has the same
    keypoint grid, and the histogram is `compute_spatial_features` over the codes of all images."""
    _extra_params = ("n_levels", "im_height", "im_width")

    def __init__(self, n_levels, im_height, im_width, feat_ext, vocabsize=None, vocab=None,
                 projection=None, backend=None):
        super(SpatialBagOfWords, self).__init__(feat_ext, vocabsize=vocabsize, vocab=vocab,
                                                projection=projection, backend=backend)
        self.n_levels = n_levels
        self.im_height = im_height
        self.im_width = im_width

    def _build_histogram(self, image_descs, codebooks):
        from .scene_functions import compute_spatial_features
        if image_descs:
            # keypoints are the same for all images; keypoint.pt is (row, col), (col, row) is needed
            cv_kps = image_descs[0].cv_keypoints
            kps = np.asarray([(k.pt if hasattr(k, "pt") else k)[::-1] for k in cv_kps])
            return compute_spatial_features(np.asarray(codebooks), kps, len(self.vocab),
                                            self.n_levels, self.im_width, self.im_height,
                                            backend=getattr(self, "_backend", None))
        return []
