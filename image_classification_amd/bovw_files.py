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
this stand-in uses the kmeans2 semantics of clustering.py.  `hist_kms` is not provided: its output
is an OpenCV matrix file whose row layout (labels, ids) is produced inside the missing binary.
"""
import os
import sys
from xml.etree import ElementTree
from xml.sax.saxutils import XMLGenerator

import numpy as np

__all__ = ["write_descriptor_file", "read_descriptor_file", "write_vocab_file", "read_vocab_file",
           "extract_from_xml", "binary_to_ndarray", "vocab_kms"]


def _write_tags(xml_file, tags):
    with open(xml_file, "w") as out:
        w = XMLGenerator(out=out, encoding="utf-8")
        w.startDocument()
        w.startElement("opencv_storage", {})
        for name, value in tags:
            w.startElement(name, {})
            w.characters(str(value))
            w.endElement(name)
        w.endElement("opencv_storage")
        w.endDocument()


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
    kinds = {"u": np.uint8, "c": np.int8, "w": np.uint16, "s": np.int16, "i": np.int32,
             "f": np.float32, "d": np.float64}
    assert dt[-1] in kinds, "unknown OpenCV element type %r" % dt
    data = np.array(node.find("data").text.split(), dtype=np.float64).astype(kinds[dt[-1]])
    return data.reshape(rows, cols, channels)


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


def main(argv=None):
    """argv of the binary: vocab_kms <in.xml> <out.xml> <vocab_size>."""
    argv = sys.argv[1:] if argv is None else argv
    if len(argv) != 3:
        sys.stderr.write("usage: vocab_kms <descriptors.xml> <vocab.xml> <vocab_size>\n")
        return 2
    vocab_kms(None, argv[0], int(argv[2]), argv[1])
    return 0


if __name__ == "__main__":
    sys.exit(main())
