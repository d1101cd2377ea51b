"""cv2-free carrier of an image's keypoints and descriptors.

Same attributes as the reference's ImageDesc (model/imagedesc.py:8-38): `cv_keypoints`,
`key_size`, `descriptors`, `desc_rows`, `desc_cols`, `keypoints` ([n, 2] array), and the same
assertion that the number of keypoints equals the number of descriptor rows.  Keypoints may be
objects with a `.pt` attribute (cv2.KeyPoint) or plain (x, y) pairs.
"""
import numpy as np

__all__ = ["ImageDesc"]


class ImageDesc(object):

    def __init__(self, keypoints, descriptors):
        self.cv_keypoints = keypoints
        self.key_size = len(keypoints)
        self.descriptors = descriptors
        if self.descriptors is not None:
            self.desc_rows = descriptors.shape[0]
            self.desc_cols = descriptors.shape[1]
        else:
            self.desc_rows = 0
            self.desc_cols = 0
        assert self.key_size == self.desc_rows, \
            "Number of keywpoint and descriptors is different"
        self._keypoints = None

    @property
    def keypoints(self):
        """[n, 2] float64 array of (x, y); built on first use (quantisation never needs it, and
        converting a million keypoint objects costs more than searching their descriptors)."""
        if self._keypoints is None:
            if self.key_size == 0:
                self._keypoints = np.array([])
            elif hasattr(self.cv_keypoints[0], "pt"):
                self._keypoints = np.array([k.pt for k in self.cv_keypoints],
                                           dtype=np.float64).reshape(self.key_size, -1)[:, :2]
            else:   # plain (x, y) pairs or an [n, 2] array: one conversion, no Python loop
                self._keypoints = np.array(self.cv_keypoints, dtype=np.float64).reshape(
                    self.key_size, -1)[:, :2]
        return self._keypoints

    @keypoints.setter
    def keypoints(self, value):
        self._keypoints = value

    def __eq__(self, other):
        if self.descriptors is None and len(self.cv_keypoints) == 0:
            return other.descriptors is None and len(other.cv_keypoints) == 0
        if self.keypoints.shape == other.keypoints.shape:
            return bool(np.allclose(self.keypoints, other.keypoints) and
                        np.allclose(self.descriptors, other.descriptors))
        return False

    def __ne__(self, other):
        return not self.__eq__(other)

    __hash__ = None
