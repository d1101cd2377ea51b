"""The `.xml` + `.xml.desc` file pair the reference exchanges with its external BoVW binaries, and a
device-backed stand-in for the `vocab_kms` binary.

File contract (all citations relative to the reference tree):
  * descriptors  `X.xml` = `<opencv_storage><cols>D</cols><nDescriptors>N</nDescriptors>
    </opencv_storage>` written with xml.sax's XMLGenerator, next to `X.xml.desc` = N*D raw
    float32, row-major (model/features.py:1343-1365).
  * vocabulary   `V.xml` with the integer tags `numCenters` and `dimension`, next to `V.xml.desc` =
    K*D raw float32 (read at model/features.py:1335-1341 and
    vision/utils/scene_functions.py:325-326).
  * integer tags are read with ElementTree from the children of the root, each exactly once
    (vision/utils/scene_functions.py:394-416); raw arrays with np.fromfile + reshape (:275-277).

`vocab_kms(bin_dir, in_file_projection, in_vocab_size, out_file)` keeps the signature and the guard
of the wrapper at vision/utils/bovw_functions.py:111-119.  The binary it wraps is not part of the
reference, so its k-means variant (initialisation, iteration count, empty-cluster rule) is unknown;
this stand-in uses the kmeans2 semantics of clustering.py.

`hist_kms(bin_dir, in_file_vocab, in_file_projection, out_file)` keeps the signature of the wrapper
at vision/utils/bovw_functions.py:122-130 (argv of the binary: vocabulary, descriptors, output).
Its source is missing too; what IS in the reference are the readers of its output
(`svm_model` / `predict`, vision/utils/bovw_functions.py:133-159): the file is one OpenCV matrix,
loaded whole with `cv.Load(path)`, one row per image =
    [label, hist_0 .. hist_{K-1}, video_id, timestamp]
(`hist[:, 0]` labels, `hist[:, 1:-2]` features, `hist[:, -2:]` metadata).  The format written
here is derived from those readers; the node name, the float32 element type, the L1 normalisation
(the rule of the only in-repo histogram, model/features.py:1178-1179) and the way image
boundaries, labels and ids reach the binary are not documented anywhere in the reference, so that
part of the contract is unpinned.  Image boundaries come from an `nImg` tag (equal image sizes,
as dense-grid extraction produces and compute_spatial_features_from_xml assumes,
vision/utils/scene_functions.py:330-334) or an `imageOffsets` matrix; labels and ids from optional
`labels` / `ids` matrices of the descriptor file, zeros otherwise.
"""
import os
import sys
from xml.etree import ElementTree
from xml.sax.saxutils import XMLGenerator

import numpy as np

__all__ = ["write_descriptor_file", "read_descriptor_file", "write_vocab_file", "read_vocab_file",
           "extract_from_xml", "binary_to_ndarray", "vocab_kms", "hist_kms", "write_opencv_matrix",
           "load_opencv_matrix", "load_first_opencv_matrix"]


def _write_tags(xml_file, tags, matrices=()):
    with open(xml_file, "w") as out:
        w = XMLGenerator(out=out, encoding="utf-8")
        w.startDocument()
        w.startElement("opencv_storage", {})
        for name, value in tags:
            w.startElement(name, {})
            w.characters(str(value))
            w.endElement(name)
        for name, mat in matrices:
            _write_matrix_node(w, name, mat)
        w.endElement("opencv_storage")
        w.endDocument()


_DT = {"u": np.uint8, "c": np.int8, "w": np.uint16, "s": np.int16, "i": np.int32,
       "f": np.float32, "d": np.float64}


def _write_matrix_node(w, name, mat):
    """One `type_id="opencv-matrix"` node: <rows>, <cols>, <dt>, whitespace-separated <data>."""
    mat = np.asarray(mat)
    if mat.ndim == 1:
        mat = mat[:, None]
    dt = [k for k, v in _DT.items() if np.dtype(v) == mat.dtype]
    if not dt:
        mat = mat.astype(np.float64)
        dt = ["d"]
    fmt = "%d" if mat.dtype.kind in "iu" else ("%.8e" if mat.dtype == np.float32 else "%.17e")
    w.startElement(name, {"type_id": "opencv-matrix"})
    for tag, value in (("rows", mat.shape[0]), ("cols", mat.shape[1]), ("dt", dt[0])):
        w.startElement(tag, {})
        w.characters(str(value))
        w.endElement(tag)
    w.startElement("data", {})
    for row in mat:
        w.characters("\n    " + " ".join(fmt % v for v in row))
    w.endElement("data")
    w.endElement(name)


def write_opencv_matrix(xml_file, name, mat, tags=()):
    """An OpenCV XML storage file holding one matrix node (and optional integer tags)."""
    _write_tags(xml_file, list(tags), [(name, mat)])
    return xml_file


def write_descriptor_file(xml_file, feats):
    """`Vocabulary._write_to_file` (model/features.py:1357-1365) for a given path."""
    feats = np.asarray(feats)
    num_desc, cols = feats.shape
    _write_tags(xml_file, [("cols", cols), ("nDescriptors", num_desc)])
    feats.astype(np.float32).tofile(xml_file + ".desc")
    return xml_file


def write_vocab_file(xml_file, centers):
    centers = np.asarray(centers)
    _write_tags(xml_file, [("numCenters", centers.shape[0]), ("dimension", centers.shape[1])])
    centers.astype(np.float32).tofile(xml_file + ".desc")
    return xml_file


def extract_from_xml(path, tags):
    """Integer content of root children, each tag exactly once (scene_functions.py:394-416)."""
    root = ElementTree.parse(path).getroot()
    ret = []
    for tag in tags:
        res = root.findall(tag)
        assert len(res) >= 1, "Not found: %s" % tag
        assert len(res) <= 1, "Multiple tags: %s" % tag
        ret.append(int(res[0].text))
    return ret


def load_opencv_matrix(path, name):
    """One `type_id="opencv-matrix"` node of an OpenCV XML storage file as an ndarray
    [rows, cols, channels] (what `np.asarray(cv.Load(path, name=name))` gives at
    vision/utils/scene_functions.py:328).  The node holds <rows>, <cols>, <dt> (element type, with
    an optional leading channel count such as "2f") and whitespace-separated <data>."""
    root = ElementTree.parse(path).getroot()
    nodes = root.findall(name)
    assert len(nodes) == 1, "Expected exactly one <%s> node, found %d" % (name, len(nodes))
    node = nodes[0]
    rows = int(node.find("rows").text)
    cols = int(node.find("cols").text)
    dt = node.find("dt").text.strip().strip('"')
    channels = int(dt[:-1]) if len(dt) > 1 else 1
    kinds = _DT
    assert dt[-1] in kinds, "unknown OpenCV element type %r" % dt
    data = np.array(node.find("data").text.split(), dtype=np.float64).astype(kinds[dt[-1]])
    return data.reshape(rows, cols, channels)


def load_first_opencv_matrix(path):
    """The first matrix node of an OpenCV XML storage file as a 2-D ndarray: what
    `np.asarray(cv.Load(path))` gives for a single-channel matrix
    (vision/utils/bovw_functions.py:134,149)."""
    root = ElementTree.parse(path).getroot()
    for node in root:
        if node.get("type_id") == "opencv-matrix":
            m = load_opencv_matrix(path, node.tag)
            return m[:, :, 0] if m.shape[2] == 1 else m
    raise AssertionError("no opencv-matrix node in %s" % path)


def binary_to_ndarray(filename, dims, dtype):
    x = np.fromfile(filename, dtype=dtype)
    return x.reshape(dims)


def read_descriptor_file(xml_file):
    dims = extract_from_xml(xml_file, ["nDescriptors", "cols"])
    return binary_to_ndarray("%s.desc" % xml_file, dims, np.float32)


def read_vocab_file(xml_file, as_float64=False):
    """float32 [K, D] as scene_functions.py:325-326 reads it; float64 as `Vocabulary._read_vocab`
    (struct.unpack into Python floats, model/features.py:1335-1341) returns it."""
    dims = extract_from_xml(xml_file, ["numCenters", "dimension"])
    v = binary_to_ndarray("%s.desc" % xml_file, dims, np.float32)
    return v.astype(np.float64) if as_float64 else v


def _verify_readable(path):
    if not (os.path.isfile(path) and os.access(path, os.R_OK)):
        raise IOError("Failed trying to read file %s" % path)
    if os.stat(path)[6] == 0:
        raise IOError("Empty file %s" % path)


def _verify_writable(path):
    if os.path.isfile(path):
        ok = os.access(path, os.W_OK)
    else:
        ok = os.access(os.path.dirname(path) or ".", os.W_OK)
    if not ok:
        raise IOError("Failed writing to file %s" % path)


def vocab_kms(bin_dir, in_file_projection, in_vocab_size, out_file, iterations=20, seed=None,
              backend=None):
    """Train a vocabulary from a descriptor file pair and write the vocabulary file pair.

    `bin_dir` is accepted for signature compatibility and ignored.
    """
    from .clustering import kmeans_fit
    _verify_readable(in_file_projection)
    _verify_writable(out_file)
    n_desc = int(ElementTree.parse(in_file_projection).find("nDescriptors").text)
    if in_vocab_size > n_desc:
        raise IOError("number of codes is larger than number of descriptors \n")
    feats = read_descriptor_file(in_file_projection)
    centers, _ = kmeans_fit(feats, int(in_vocab_size), iters=iterations, minit="points", rng=seed,
                            backend=backend)
    write_vocab_file(out_file, centers)
    return out_file


def _image_offsets(xml_file, n_desc):
    """Image boundaries of a descriptor file: `imageOffsets` matrix, else `nImg` equal images,
    else one image."""
    root = ElementTree.parse(xml_file).getroot()
    if root.find("imageOffsets") is not None:
        off = load_opencv_matrix(xml_file, "imageOffsets").reshape(-1).astype(np.int64)
        if off[0] != 0 or off[-1] != n_desc or np.any(np.diff(off) < 0):
            raise IOError("imageOffsets of %s must rise from 0 to nDescriptors" % xml_file)
        return off
    n_img = int(root.find("nImg").text) if root.find("nImg") is not None else 1
    if n_img < 1 or n_desc % n_img:
        raise IOError("%d descriptors do not divide into %d images" % (n_desc, n_img))
    return np.arange(n_img + 1, dtype=np.int64) * (n_desc // n_img)


def hist_kms(bin_dir, in_file_vocab, in_file_projection, out_file, norm="l1", backend=None):
    """Assign every descriptor of the projection file to its nearest vocabulary word and write
    one histogram row per image: [label, K bins, video_id, timestamp] (see the module docstring).

    `bin_dir` is accepted for signature compatibility and ignored.
    """
    from .backend import default_backend
    for f in (in_file_vocab, in_file_projection):
        _verify_readable(f)
    _verify_writable(out_file)
    vocab = read_vocab_file(in_file_vocab)
    feats = read_descriptor_file(in_file_projection)
    if feats.shape[1] != vocab.shape[1]:
        raise IOError("descriptors have %d columns, the vocabulary %d"
                      % (feats.shape[1], vocab.shape[1]))
    offsets = _image_offsets(in_file_projection, feats.shape[0])
    n_img = len(offsets) - 1
    b = backend or default_backend()
    if feats.shape[0]:
        codes, _ = b.assign_top1_np(feats, b.codebook(vocab), want_dist=False)
    else:
        codes = b.i32(np.empty(0, dtype=np.int32))
    hists = b.to_host(b.hist(codes, b.i64(offsets), vocab.shape[0], norm))
    root = ElementTree.parse(in_file_projection).getroot()
    labels = np.zeros(n_img)
    ids = np.zeros((n_img, 2))
    if root.find("labels") is not None:
        labels = load_opencv_matrix(in_file_projection, "labels").reshape(-1).astype(np.float64)
    if root.find("ids") is not None:
        ids = load_opencv_matrix(in_file_projection, "ids").reshape(n_img, 2).astype(np.float64)
    if labels.shape[0] != n_img:
        raise IOError("%d labels for %d images" % (labels.shape[0], n_img))
    rows = np.concatenate([labels[:, None], np.asarray(hists, dtype=np.float64), ids], axis=1)
    write_opencv_matrix(out_file, "hist", rows.astype(np.float32))
    return out_file


def main(argv=None):
    """argv of the binaries: [vocab_kms] <in.xml> <out.xml> <vocab_size>, or
    hist_kms <vocab.xml> <descriptors.xml> <out.xml>."""
    argv = sys.argv[1:] if argv is None else list(argv)
    if argv and argv[0] == "hist_kms":
        if len(argv) != 4:
            sys.stderr.write("usage: hist_kms <vocab.xml> <descriptors.xml> <hist.xml>\n")
            return 2
        hist_kms(None, argv[1], argv[2], argv[3])
        return 0
    if argv and argv[0] == "vocab_kms":
        argv = argv[1:]
    if len(argv) != 3:
        sys.stderr.write("usage: vocab_kms <descriptors.xml> <vocab.xml> <vocab_size>\n")
        return 2
    vocab_kms(None, argv[0], int(argv[2]), argv[1])
    return 0


if __name__ == "__main__":
    sys.exit(main())
